"""Times the focal attention forward / backward alone at the metric shape (diagnostics)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from fvta_memexqa_amd import ops
N, K, T, JQ, w = 64, 6, 1200, 30, 1024  # the metric shape: K = 5 text streams + photo stream, T = 40 photos x 30
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(N, K, T, w, device="cuda", generator=g) * 0.5
q = torch.randn(N, JQ, w, device="cuda", generator=g) * 0.5
W = torch.randn(2 * w, device="cuda", generator=g) * 0.1
b = torch.zeros(1, device="cuda")
hm = torch.ones(N, K, T, dtype=torch.uint8, device="cuda"); hm[:, 5, 40:] = 0
qm = torch.ones(N, JQ, dtype=torch.uint8, device="cuda")
op = ops.FocalAttention(N, K, T, JQ, w, 2, True)
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def main_kernel_ms(n=20):  # the library's HIP-event bracket around the forward main kernel (FVTA_PROF_ATTN_FWD_MAIN = 4)
    import ctypes
    from fvta_memexqa_amd import _lib
    lib = _lib.load() if hasattr(_lib, "load") else _lib.lib
    lib.fvta_profile_enable(1)
    for _ in range(n): op.forward(h, q, hm, qm, W, b)
    torch.cuda.synchronize(); lib.fvta_profile_enable(0)
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    lib.fvta_profile_collect(4, ctypes.byref(ms), ctypes.byref(cnt))
    return ms.value / max(cnt.value, 1)
print("attn fwd ms", round(timeit(lambda: op.forward(h, q, hm, qm, W, b)), 3), "main kernel ms", round(main_kernel_ms(), 4), "dbg", os.environ.get("FVTA_DEBUG_SKIP"))
if len(sys.argv) > 1:
    gout = torch.randn(N, w, device="cuda", generator=g)
    dh = torch.zeros_like(h); dq = torch.zeros_like(q); dW = torch.zeros_like(W); db = torch.zeros(1, device="cuda")
    # accumulate = 2: the mode the Model runs (d_hq accumulated, only the valid rows of d_hinfo written, never read);
    # mode 1 additionally READS every d_hinfo row -- which is what doubled the fetch figure of profiles/r01d_attention_pmc.json
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    print("attn bwd ms (accumulate=%d)" % mode, round(timeit(lambda: op.backward(h, q, hm, qm, W, b, gout, dh, dq, dW, db, mode)), 3))
    import ctypes
    from fvta_memexqa_amd import _lib
    lib = _lib.load()
    lib.fvta_profile_enable(1)
    for _ in range(10): op.backward(h, q, hm, qm, W, b, gout, dh, dq, dW, db, mode)
    torch.cuda.synchronize(); lib.fvta_profile_enable(0)
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    lib.fvta_profile_collect(5, ctypes.byref(ms), ctypes.byref(cnt))
    print("attn_bwd_main kernel ms", round(ms.value / max(cnt.value, 1), 4))
