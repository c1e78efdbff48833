#!/usr/bin/env python3
"""Benchmark of the FVTA hot path on MI355X: QA-pairs/s for one training step
(forward + backward + optimiser update) at BASELINE.json's metric shape
(batch 64 per GPU, 40 photos x 5 text streams x 30 tokens, hidden 512).

  python bench.py --gpus N --steps K --warmup W          (N=1; N>1 without a launcher: the ranks are spawned here)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU; albums/questions are sharded over ranks (weak scaling: 64 QA
pairs per GPU), the only collective is the all-reduce of the flat gradient
buffer.  Inputs (synthetic, seeded) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  `roofline` is for the DOMINANT kernel of the step
(largest `kernel_ms_per_step` entry; today the bi-LSTM backward step), every other
bracketed kernel has its own `roofline_<name>` object.  Kernel durations come from
HIP events recorded on the launch stream inside the library (fvta_profile_*), live,
over a second pass of the same steps right after the timed region.
`cpu_baseline` times the CPU oracle ("port") on a bounded sample of the same
workload on this box's host cores (rank 0, N=1 only).  After the headline, the
default N=1 run also times the other BASELINE.json configurations that fit one GPU
(short runs, attached under `also`): configs[1] forward only (fp32 and bf16 engines),
the 1e-4-parity `bf16x3` train step, ragged lengths, configs[4] long album, the
token-id entry, and the published flag set (time warp type 5, char_emb_size 100).
"""
import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3        # fp32 MFMA (v_mfma_f32_32x32x2_f32)
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA

_T0 = time.perf_counter()


def log(msg):
    if os.environ.get("FVTA_BENCH_VERBOSE", "1") != "0":
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the clock / power governor needs ~0.1 s of load after idle to settle -- 300 headline steps in a row measured
    # 12.05, 11.89, 11.83, 11.84, 11.75 ms for the first five timed steps (behind five warm-up steps), 11.65-11.72 for the
    # other 295 (profiles/r05_bench_300_steps.json): ten warm-up steps time the steady state a training run lives in
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="metric", choices=["metric", "plumbing", "long_album"])
    ap.add_argument("--variant", default="dense", choices=["dense", "ragged"],
                    help="dense: every length = max (no padding to skip; the headline). ragged: SURVEY 8d length distribution")
    ap.add_argument("--precision", default="bf16", choices=["f32", "bf16", "bf16x3"],
                    help="bf16: BASELINE.json configs[2] (bf16 MFMA operands in the bi-LSTM, fp32 accumulate, fp32 attention); "
                         "f32: exact-fp32 engine (the 1e-4 parity path); bf16x3: the bi-LSTM's operands split in two bf16 terms, "
                         "three MFMA products per GEMM -- 1e-4 parity on the bf16 matrix pipe")
    ap.add_argument("--optimizer", default="adam", choices=["adam", "adadelta"])
    ap.add_argument("--batch", type=int, default=None, help="QA pairs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--side-priority", type=int, default=None,
                    help="priority of the photo cell's side stream (default: the highest the device offers; 0 = normal)")
    ap.add_argument("--serial-photo-forward", type=int, default=None, help="1 / 0: the photo cell's forward in front of / beside the text cell")
    ap.add_argument("--cpu-sample", type=int, default=4, help="QA pairs in the CPU-baseline sample (4: ~10-15 s of CPU work)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch-CPU threads of the baseline; 16 is the fastest setting measured on the 2x EPYC 9575F host "
                         "(64 threads: 3.5x slower, 128: 9.5x slower for this op mix; DESIGN.md section 5)")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE.json configs[1] (inference) instead of the train step")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank gets the config's batch (64 QA pairs per GPU; the driver's default run). "
                         "strong: --global-batch QA pairs in total, global/N per rank (north_star's strong-scaling target)")
    ap.add_argument("--global-batch", type=int, default=512, help="--scaling strong: total QA pairs (BASELINE.json configs[3]: 512)")
    ap.add_argument("--graph", default="fvta", choices=["fvta", "model_py"],
                    help="fvta: model_v2.py, the FVTA model (BASELINE.json's metric); model_py: model.py's soft-attention "
                         "baselines with every attention on (use_ml_att, use_mm_att, use_direct_links, use_choices_att, "
                         "use_question_att) -- a side measurement, same shape and encoders")
    ap.add_argument("--front-end", action="store_true",
                    help="enter with the reference's token-id feed (SURVEY 8f rank 1): the char-CNN / word / photo embedding "
                         "front-end and its gradients are inside the timed step (the headline enters at the encoder inputs)")
    ap.add_argument("--char-emb-size", type=int, default=8,
                    help="--front-end only: char embedding width (8: the reference's default; 100: the published flag set, "
                         "README.MD:144 -- a 500-deep char-CNN window, run on the fp32 MFMA engine)")
    ap.add_argument("--time-warp", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="--use_time_warp --warp_type N (model_v2.py:953-1009); 5 is the published FVTA flag set (README.MD:144-147)")
    ap.add_argument("--also", default="auto", choices=["auto", "on", "off"],
                    help="after the headline, time the other BASELINE.json configurations that fit one GPU in short runs and "
                         "attach them as `also` (auto: for the default N=1 headline only)")
    ap.add_argument("--gc", default="freeze", choices=["freeze", "default"],
                    help="freeze: gc.collect(); gc.freeze(); gc.disable() once the model and the batch are built, as a training "
                         "loop that owns its process does (a generation-2 pass over torch's and numpy's heaps is a pause of tens of "
                         "milliseconds on the host); default: CPython's collector left alone (diagnosis)")
    ap.add_argument("--also-helper", action="store_true", help=argparse.SUPPRESS)   # internal: the `also` orchestrator (no GPU)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch --
def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks as fresh child processes -- the parent has not
    touched the GPU -- with the rendezvous environment torch.distributed.run would give them, and exit with the worst
    of their codes.  Only rank 0 prints the JSON line (the children share this process's stdout)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:              # a rank that dies takes the others with it (they would wait in the rendezvous otherwise)
        time.sleep(0.2)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                if p.returncode and not rc:
                    rc = p.returncode
                    for q in live:
                        q.terminate()
    if rc:
        print("bench.py: a rank of the self-launched %d-GPU run failed (exit code %d); no line was printed for %d GPUs" %
              (args.gpus, rc, args.gpus), file=sys.stderr, flush=True)
    sys.exit(rc if 0 <= rc < 256 else 1)


# ---------------------------------------------------------------------------------------------- cpu baseline --
def _median_time(fn, passes):
    """1 warm-up + `passes` timed calls, median seconds"""
    times = []
    for it in range(passes + 1):
        t0 = time.perf_counter()
        fn()
        if it:
            times.append(time.perf_counter() - t0)
    return statistics.median(times)


def cpu_baseline(spec_kw, sample_n, forward_only, threads, literal_n=1):
    """The reference's CPU path cannot run (Python-2 / TF-1); what is timed beside the GPU is the CPU oracle, on a
    bounded sample of the same workload on this box's host cores:
      fused   oracle/fvta_fused.py, torch-CPU fp32: forward+backward (the metric's unit) and forward only;
      literal oracle/fvta_literal.py, NumPy fp32, the op-for-op mirror of the TF graph (tile/concat/linear attention
              materialised per (n,k), per-stream LSTM loops with concat([x,h]) . kernel, pad+stack): forward only --
              it has no autograd.
    `value` is the fused forward+backward rate (forward-only runs: the faster of the two forward rates)."""
    import numpy as np
    import torch
    from fvta_memexqa_amd.synth import SynthSpec, make_inputs, make_params, to_numpy
    from oracle import fvta_fused as F
    from oracle import fvta_literal as Lit
    torch.set_num_threads(max(1, min(threads, host_cores())))
    spec = SynthSpec(**dict(spec_kw, N=sample_n))
    params = {k: v.requires_grad_(True) for k, v in make_params(spec).items()}
    inputs = make_inputs(spec)

    def fwd_bwd():
        for p in params.values():
            p.grad = None
        F.fvta_forward(params, inputs, spec.cfg())["loss"].backward()

    def fwd():
        with torch.no_grad():
            F.fvta_forward(params, inputs, spec.cfg())

    t_fb = None if forward_only else _median_time(fwd_bwd, 3)
    t_f = _median_time(fwd, 2 if not forward_only else 3)
    out = dict(unit="QA-pairs/s", cores=torch.get_num_threads(), kind="port",
               fused=dict(fwd_only=round(sample_n / t_f, 3), sample_qa_pairs=sample_n,
                          note="oracle/fvta_fused.py, torch-CPU fp32, %d threads, median (1 warm-up)" % torch.get_num_threads()))
    if t_fb is not None:
        out["fused"]["fwd_bwd"] = round(sample_n / t_fb, 3)
    # the literal mirror: 1 warm-up + 3 timed passes, median (a pass is ~1.6 s per QA pair at the metric shape)
    lspec = SynthSpec(**dict(spec_kw, N=literal_n))
    lp = to_numpy(make_params(lspec), np.float32)
    li = to_numpy(make_inputs(lspec), np.float32)
    t_l = _median_time(lambda: Lit.fvta_forward(lp, li, lspec.cfg()), 3)
    out["literal"] = dict(fwd_only=round(literal_n / t_l, 3), sample_qa_pairs=literal_n,
                          note="oracle/fvta_literal.py, NumPy fp32 op-for-op mirror, forward only (no autograd), 1 warm-up + 3 timed passes, median")
    out["faster_forward"] = "fused" if out["fused"]["fwd_only"] >= out["literal"]["fwd_only"] else "literal"
    if forward_only:
        out["value"] = max(out["fused"]["fwd_only"], out["literal"]["fwd_only"])
    else:
        out["value"] = out["fused"]["fwd_bwd"]
    out["sample"] = ("%d QA pairs of the same shape through the fused torch-CPU oracle (%s, median of 3, %.2f s each); "
                     "%d QA pair through the NumPy literal mirror (forward only, median of 3, %.1f s each)"
                     % (sample_n, "forward only" if forward_only else "forward+backward", t_f if forward_only else t_fb,
                        literal_n, t_l))
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        out["cpu_model"], out["host_logical_cpus"] = (model[0] if model else "?"), host_cores()
    except Exception:
        pass
    return out


# --------------------------------------------------------------------------------------------------- one case --
GC_MODE = "default"
PROF_IDS = [("lstm_step_fwd", 1), ("lstm_step_bwd", 2), ("lstm_dw", 3), ("attn_fwd_main", 4), ("attn_bwd_main", 5),
            ("lstm_dx", 6), ("lstm_step_fwd_photo_cell", 17)]


def pmc_traffic(fname, key):
    """(HBM bytes per launch, provenance) from a committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs
    at this same shape; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  Not collected live: counters need
    their own profiler pass -- so the note says WHEN and on WHICH commit the file was collected (profiles/pmc_meta.json),
    and a kernel that has changed since shows as a stale commit there, not as a silently wrong number."""
    try:
        d = json.load(open(os.path.join(HERE, "profiles", fname)))
        meta = {}
        try:
            meta = json.load(open(os.path.join(HERE, "profiles", "pmc_meta.json"))).get(fname, {})
        except Exception:
            pass
        note = "profiles/%s@%s" % (fname, meta.get("commit", "?"))
        for k, v in d.items():
            if key in k:
                return v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"], note
    except Exception:
        pass
    return None, None


def run_case(args, lib, ws, rank, local, probe_gbs=None):
    """Builds the model of one configuration, runs warm-up + the timed region + the bracketed pass; returns
    (result dict for rank 0 or None, spec keywords for the CPU baseline)."""
    import torch
    from fvta_memexqa_amd import dist
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
    from fvta_memexqa_amd.trainer import Trainer
    dev = torch.device("cuda", local)
    kw = dict(CONFIGS[args.config], dense=(args.variant == "dense"))
    if args.batch:
        kw["N"] = args.batch
    if args.scaling == "strong":      # fixed total work: global/ws QA pairs per rank
        lo, hi = dist.shard_range(args.global_batch, ws, rank)
        kw["N"] = hi - lo
    spec = SynthSpec(**kw)
    cfg = dict(spec.cfg(), batch_size=spec.N, precision=args.precision, optimizer=args.optimizer,
               init_lr=0.001 if args.optimizer == "adam" else 0.5, gc_freeze=(args.gc == "freeze"))
    if args.time_warp:
        cfg.update(use_time_warp=True, warp_type=args.time_warp)
    if args.side_priority is not None:
        cfg["side_stream_priority"] = args.side_priority
    if args.serial_photo_forward is not None:
        cfg["serial_photo_forward"] = bool(args.serial_photo_forward)
    if args.front_end:   # README.MD:144-147 sizes: 100-d GloVe + 100-d char-CNN, 2537-d photo features -> 100
        from fvta_memexqa_amd.synth import make_token_inputs
        cfg.update(word_vocab_size=400, word_emb_size=100, use_char=True, char_vocab_size=100, max_word_size=16,
                   char_emb_size=args.char_emb_size, char_out_size=100, image_feat_dim=2537, use_image_trans=True, image_trans_dim=100)
    if args.graph == "model_py":
        from fvta_memexqa_amd.model import Model as ModelPy
        probe = make_inputs(SynthSpec(**dict(kw, N=1, P=1, L=2)))
        cfg.update(add_tanh=False, use_ml_att=True, use_mm_att=True, use_direct_links=True, use_choices_att=True,
                   use_question_att=True, simiMatrix=1, ctx_streams=ModelPy.streams_of(probe))
        model = ModelPy(cfg, text_in=spec.text_in, img_in=spec.img_in, device=dev)
    else:
        model = Model(cfg, text_in=spec.text_in, img_in=spec.img_in, device=dev)
    trainer = Trainer(model, cfg)
    trainer.need_dx = True   # the real model trains its embeddings: gradients flow into the encoder inputs
    log('model built; generating synthetic inputs')
    if args.front_end:
        model.set_existing_emb(torch.randn(20000, 100, generator=torch.Generator().manual_seed(5)) * 0.5)
        inputs = make_token_inputs(spec, VW=400, VF=20000, VC=100, W=16, rank=rank)
        inputs["image_emb_mat"] = torch.randn(inputs["n_image_rows"], 2537, generator=torch.Generator().manual_seed(6))
    else:
        inputs = make_inputs(spec, rank=rank)                  # synthetic, seed 1234+rank
    L = model.load_inputs(inputs, training=not args.forward_only)   # resident in HBM from here on
    del inputs
    log('inputs resident in HBM; warm-up')

    def step():
        if args.forward_only:
            model.forward(L)
        else:
            trainer.step_device(L)

    global GC_MODE
    GC_MODE = args.gc
    import gc
    # the collector is the TRAINER's business (Trainer.own_host, config `gc_freeze`, default on: what every user of
    # Trainer.step / step_device gets); --gc default switches it off for the comparison case `also.headline_gc_default`
    if args.gc == "freeze" and args.forward_only:
        trainer.own_host()       # (the forward-only cases never call step_device)
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log('warm-up step %d done' % i)
    # ---- the timed region: exactly `steps` steps between barrier + synchronize pairs, no profiling brackets inside;
    # one HIP event per step on the main stream (every side stream has joined it when a step ends) gives the
    # per-step device times whose median is reported beside the wall-clock mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for m in marks:          # (an event's HIP object is created by its first record: not inside the timed region)
        m.record()
    # the host side of the timed region is watched too: the collector's pauses (gc.callbacks), the process's page faults
    # and context switches (getrusage), and the host's enqueue time of every single step
    import gc
    import resource
    gc_log = []

    def gc_watch(phase, info, _t=[0.0]):
        if phase == "start":
            _t[0] = time.perf_counter()
        else:
            gc_log.append((info.get("generation"), round((time.perf_counter() - _t[0]) * 1e3, 3), round((_t[0] - t0) * 1e3, 3)))

    host_t = [0.0] * (args.steps + 1)
    dist.barrier()
    torch.cuda.synchronize()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    gc.callbacks.append(gc_watch)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
        host_t[i + 1] = time.perf_counter() - t0
    host_enqueue = time.perf_counter() - t0      # the host's share: when it equals the step time the run is launch bound
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.callbacks.remove(gc_watch)
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_step_ms = [round((host_t[i + 1] - host_t[i]) * 1e3, 3) for i in range(args.steps)]
    rank_elapsed = dist.gather_over_ranks(elapsed, dev)      # every rank's own wall clock over the timed region
    elapsed = dist.max_over_ranks(elapsed, dev)
    pg = dist.describe()
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    log('timed region done: %.3f s for %d steps (per-step HIP events: median %.3f ms)'
        % (elapsed, args.steps, statistics.median(step_ms)))
    # ---- a second pass of the same steps with the library's HIP-event brackets around the kernels the rooflines
    # are quoted for (events on the launch streams); kept out of the headline
    lib.fvta_profile_enable(1)
    lib.fvta_lstm_bwd_kernel_counts((ctypes.c_int64 * 3)())      # (reading resets the counters)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    lib.fvta_profile_enable(0)
    dist.barrier()

    def collect(pid):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        lib.fvta_profile_collect(pid, ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value

    prof = {name: collect(pid) for name, pid in PROF_IDS}
    bwd_counts = (ctypes.c_int64 * 3)()
    lib.fvta_lstm_bwd_kernel_counts(bwd_counts)     # text-cell backward-step launches by kernel since the bracketed pass began
    bwd_counts = dict(zip(("lstm_bwd_fused_bf16", "lstm_bwd_ring_bf16", "lstm_bwd_wreg_bf16"), (int(v) for v in bwd_counts)))
    if rank != 0:
        return None, kw
    total_qa = spec.N * ws * args.steps
    value = total_qa / elapsed
    # ---- rooflines.  The bi-LSTM step kernels of the text cell move, per active (row, direction) and launch:
    #   forward   x shadow in_i*2 B + h shadow read d*2 + c_prev d*4 + c d*4 + h (fp32, into the context tensor) d*4
    #             + h shadow d*2 + saved gates 4d*2, and do 2*(in+d)*4d flops                             
    #   backward  saved bf16 gates 8 B + c_{t-1} 4 + d_out 4 + dc 4 + 4 + dz 8 = 32 B per unit (the re-read of dz(t+1),
    #             8 B, is not counted)
    #   dx        dz read once (8 B per unit) + the input gradient written;   dW   2*(in+d)*4d flops per (row, direction)
    T = L.groups["text"]
    calls = args.steps
    lens = T.lens.float()
    dp = model.dp
    in_i = ((spec.text_in + 1 + 31) // 32) * 32
    fl_text = float((2 * (lens * 2.0 * (spec.text_in + dp) * 4 * dp - (lens > 0).float() * 2.0 * dp * 4 * dp)).sum().item())
    # per row and direction: x, h(t-1) read (bf16); c(t-1) read, c(t) written (fp32); h(t) written as the bf16 shadow row, and
    # as fp32 for the rows whose readers take fp32 (under shadow rows: the question / choice sequences only); bf16 gates
    shadow = bool(getattr(L, "shadow", False))
    h32 = torch.ones_like(lens)
    if shadow:
        h32[T.segs[0]["count"] + T.segs[1]["count"]:] = 0.0      # (the text cell's sequences: q, choices, then the context streams)
    x3 = args.precision == "bf16x3"      # the split engine: two stored bf16 terms per operand value, fp32 saved gates
    if x3:   # x, h(t-1) read as (hi, lo); c read + written; h(t) written as (hi, lo) [+ fp32]; fp32 gates 16 B per unit
        by_fwd = float((2 * lens * (in_i * 4 + dp * 4 + dp * 4 * 2 + dp * 4 + 4 * dp * 4) + 2 * lens * h32 * dp * 4).sum().item())
        by_bwd = float((2 * lens * dp * 48).sum().item())      # gates 16, c 4, d_out 4, dc 4 + 4, dz (hi, lo) 16
        by_dx = float((2 * lens * dp * 16 + lens * spec.text_in * 4).sum().item())
    else:
        by_fwd = float((2 * lens * (in_i * 2 + dp * 2 + dp * 4 * 2 + dp * 2 + 4 * dp * 2) + 2 * lens * h32 * dp * 4).sum().item())
        by_bwd = float((2 * lens * dp * 32).sum().item())
        by_dx = float((2 * lens * dp * 8 + lens * spec.text_in * 4).sum().item())
    is_bf = args.precision in ("bf16", "bf16x3")
    # the matrix pipe's peak for this engine, and the hardware flops per algorithmic flop (the split engine runs three
    # bf16 products per logical one)
    peak_tf, hw_mult = (PEAK_BF16_TFLOPS, 3.0 if args.precision == "bf16x3" else 1.0) if is_bf else (PEAK_F32_TFLOPS, 1.0)
    wreg_fwd = ((args.precision == "bf16" and (dp, in_i) in ((512, 224), (512, 128), (1024, 224), (1024, 128), (128, 128), (128, 32)))
                or (x3 and (dp, in_i) in ((512, 224), (512, 128), (128, 128), (128, 32))))
    # which kernel ran the backward step is the LIBRARY's word (fvta_lstm_bwd_kernel_counts), not re-derived here
    bwd_main = max(bwd_counts, key=lambda k: bwd_counts[k]) if sum(bwd_counts.values()) else "lstm_bwd_fused_bf16"
    ring_bwd = bwd_main == "lstm_bwd_ring_bf16"
    dense_metric = (args.config == "metric" and args.variant == "dense" and not args.batch and args.graph == "fvta"
                    and args.precision in ("bf16", "bf16x3") and not args.time_warp and not args.front_end)
    roofs = {}

    def hbm_roof(name, kernel, prof_key, bytes_per_call, flops_per_call=None, note=None, traffic=None):
        ms, n = prof[prof_key]
        if not n:
            return
        gbs = bytes_per_call * calls / (ms * 1e-3) / 1e9
        r = dict(kernel=kernel, bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                 traffic=None, algorithmic_bytes_per_call=bytes_per_call, ms_per_step=round(ms / args.steps, 4), launches=int(n),
                 avg_launch_ms=round(ms / n, 4))
        if flops_per_call:
            tf = flops_per_call * calls / (ms * 1e-3) / 1e12
            r.update(mfma_tflops=round(tf, 1), mfma_frac=round(hw_mult * tf / peak_tf, 4))
        if note:
            r["note"] = note
        if traffic and dense_metric:
            r["traffic"], r["traffic_note"] = pmc_traffic(*traffic)
        if probe_gbs:
            r["achievable_peak"] = probe_gbs
            r["frac_of_achievable"] = round(gbs / probe_gbs, 4)
        roofs[name] = r

    def mfma_roof(name, kernel, prof_key, flops_per_call, note=None):
        ms, n = prof[prof_key]
        if not n:
            return
        tf = flops_per_call * calls / (ms * 1e-3) / 1e12
        roofs[name] = dict(kernel=kernel, bound="mfma", achieved=round(hw_mult * tf, 1), peak=peak_tf, unit="TFLOP/s",
                           frac=round(hw_mult * tf / peak_tf, 4), traffic=None, algorithmic_flops_per_call=flops_per_call,
                           ms_per_step=round(ms / args.steps, 4), launches=int(n), avg_launch_ms=round(ms / n, 4))
        if hw_mult != 1.0:
            roofs[name]["note"] = "hardware flops = %.0f x the algorithmic flops (split-bf16 products), against the bf16 peak" % hw_mult
        if note:
            roofs[name]["note"] = note

    if args.precision == "f32":   # exact-fp32 MFMA: 64 flop/clk/SIMD -- the matrix pipe bounds this engine, not HBM
        mfma_roof("lstm_fwd", "lstm_step_fwd_f32", "lstm_step_fwd", fl_text)
    else:
        hbm_roof("lstm_fwd", "lstm_fwd_wreg_bf16 (forward step, weights in registers)" if wreg_fwd else "lstm_step_fwd_bf16",
                 "lstm_step_fwd", by_fwd, fl_text, traffic=(("r06_lstm_bf16x3_pmc.json", "lstm_fwd_wreg_bf16<fvta::WregCfg<14, 32, 1, true") if x3 else ("r06_lstm_pmc.json", "lstm_fwd_wreg_bf16<fvta::WregCfg<14, 32, 2,")))
        ms_p, n_p = prof["lstm_step_fwd_photo_cell"]
        if n_p and "lstm_fwd" in roofs:
            # rocprofv3 --stats averages per kernel SYMBOL: the photo cell (side stream, small launches) runs the same
            # symbol, so its launches are reported too -- the all-launch mean is the figure to hold against AverageNs
            ms_f, n_f = prof["lstm_step_fwd"]
            roofs["lstm_fwd"].update(launches_photo_cell=int(n_p), avg_launch_ms_photo_cell=round(ms_p / n_p, 4),
                                     avg_launch_ms_all_launches=round((ms_f + ms_p) / (n_f + n_p), 4))
    if not args.forward_only and is_bf:
        hbm_roof("lstm_bwd", {"lstm_bwd_ring_bf16": "lstm_bwd_ring_bf16 (backward step, weights in registers, pipelined)",
                              "lstm_bwd_wreg_bf16": "lstm_bwd_wreg_bf16 (backward step, weights in registers)",
                              "lstm_bwd_fused_bf16": "lstm_bwd_fused_bf16 (backward step: dz(t+1) Wh^T + gate gradient)"}[bwd_main]
                 + "; text-cell launches by kernel in the bracketed pass: %r" % bwd_counts,
                 "lstm_step_bwd", by_bwd, fl_text * dp / (spec.text_in + dp),
                 note="algorithmic bytes: the gate gradient (32 B per row and unit: gates 8, c 4, d_out 4, dc 4 + 4, dz 8); the "
                      "re-read of dz(t+1) (8 B) is not counted; dx has its own bracket",
                 traffic=(("r06_lstm_bf16x3_bwd_pmc.json", "lstm_bwd_fused_bf16<2, 4, 2,") if x3 else ("r06_lstm_bwd_pmc.json", "lstm_bwd_ring_bf16" if ring_bwd else "lstm_bwd_fused_bf16<2, 4, 1,")))
        hbm_roof("lstm_dx", "lstm_dx_bf16 (input gradient of all steps, both directions in one launch)", "lstm_dx", by_dx,
                 fl_text * spec.text_in / (spec.text_in + dp), traffic=(("r06_lstm_bf16x3_bwd_pmc.json", "lstm_dx_bf16<2, 64, 2, true, true, 2") if x3 else ("r06_lstm_bwd_pmc.json", "lstm_dx_bf16<2, 64, 2, true, true, 1")))
        mfma_roof("lstm_dw", "lstm_dw_bf16 (weight gradient: [x|h|1]^T dz over every row and step)", "lstm_dw", fl_text)
    # ---- attention kernels against HBM: algorithmic bytes = valid rows * w * 4 + question + output; the
    # backward reads the rows and writes their gradient
    valid_rows = int(L.hall_mask.sum().item())
    row_b = 2.0 if shadow else 4.0     # shadow rows: the attention reads the encoders' bf16 rows
    att_bytes = valid_rows * model.wp * row_b + (spec.N * L.JQ * model.wp + spec.N * model.wp) * 4.0
    if args.graph == "fvta":   # (model.py's graph runs seven large 1-D attentions under the same bracket)
        kname = "attn_fwd_pair16h (bf16 shadow rows)" if shadow else (("attn_fwd_pair16" if model.wp >= 512 and model.simi != 4 and os.environ.get("FVTA_ATTN_WAVE16", "3") in ("2", "3")
                  else "attn_fwd_rows16") if (L.JQ <= 32 and 128 <= model.wp <= 1024) else
                 ("attn_fwd_wide" if model.wp == 2048 and L.JQ <= 64 and model.simi != 4 and os.environ.get("FVTA_ATTN_WAVE16", "3") != "0"
                  else "attn_fwd_main"))
        # (the library's profile bracket covers the context attention only, not the K = 1 question attention)
        hbm_roof("attention", kname + " (fvta_attn_fwd main kernel)", "attn_fwd_main", att_bytes,
                 traffic=("r06_attention_pmc.json", "attn_fwd_pair16h" if shadow else "attn_fwd_pair16<"))
        if not args.forward_only:
            hbm_roof("attention_bwd", "attn_bwd_main", "attn_bwd_main", valid_rows * model.wp * (row_b + 4.0),
                     traffic=("r06_attention_pmc.json", "attn_bwd_main<256, 1, 32, false, false, true>" if shadow else "attn_bwd_main<256, 1, 32, false, false, false>"))
    # the DOMINANT kernel of the step (largest bracketed time per step) is `roofline`; the rest are roofline_<name>
    kms = {k: round(v[0] / args.steps, 4) for k, v in prof.items()}
    order = sorted(roofs, key=lambda k: -roofs[k]["ms_per_step"])
    for r in roofs.values():         # the line stays short enough for the driver's record: the notes live in DESIGN.md section 5
        r.pop("note", None)

    def stats(v):
        return dict(min=round(min(v), 3), median=round(statistics.median(v), 3), max=round(max(v), 3), argmax=v.index(max(v)))

    out = dict(
        metric="QA-pairs/sec (fwd+bwd) at B=64, 40 photos x 5 streams x 30 tok, h=512" if args.config == "metric" and not args.forward_only
        and args.graph == "fvta" else "QA-pairs/sec (model.py graph, %s, config %s)" % ("fwd" if args.forward_only else "fwd+bwd", args.config)
        if args.graph == "model_py" else "QA-pairs/sec (%s, config %s)" % ("fwd" if args.forward_only else "fwd+bwd", args.config),
        value=round(value, 2), unit="QA-pairs/s", n_gpus=ws, steps=args.steps, warmup=args.warmup,
        ms_per_step=round(elapsed / args.steps * 1e3, 3), ms_per_step_event_median=round(statistics.median(step_ms), 3),
        higher_is_better=True, scaling=args.scaling, vs_baseline=None,
        dtype=args.precision, data="synthetic",
        config=dict(workload=("BASELINE.json configs[2] train step (fwd+bwd+%s)" % args.optimizer if not args.forward_only
                              else "BASELINE.json configs[1] forward only") + ", shape '%s', %s lengths" % (args.config, args.variant)
                    + (", token-id entry (embedding front-end inside the step, char_emb_size %d)" % args.char_emb_size
                       if args.front_end else "")
                    + (", --use_time_warp --warp_type %d" % args.time_warp if args.time_warp else "")
                    + (" -- model.py's soft-attention baseline graph" if args.graph == "model_py" else ""),
                    qa_pairs_per_gpu=spec.N, albums=spec.A, photos=spec.P, text_streams=spec.S, tokens=spec.L, hidden=spec.d,
                    global_batch=spec.N * ws, K=L.K, T=L.T, JQ=L.JQ,
                    parallelism="dp%d (QA pairs sharded, flat-gradient all-reduce)" % ws),
        roofline=roofs[order[0]] if order else None,
        roofline_is="roofline_%s" % order[0] if order else None,
    )
    out["host_enqueue_ms_per_step"] = round(host_enqueue / args.steps * 1e3, 3)
    # the evidence behind ms_per_step: min / median / max (and where the maximum fell) of every step's device time (HIP events
    # on the main stream) and host enqueue time, and what the host did meanwhile
    out["step_ms"] = stats([round(v, 3) for v in step_ms])
    out["host_step_ms"] = stats(host_step_ms)
    out["host_watch"] = dict(gc_mode=GC_MODE, gc_enabled=gc.isenabled(), gc_collections=gc_log[:8],
                             minor_faults=ru1.ru_minflt - ru0.ru_minflt, major_faults=ru1.ru_majflt - ru0.ru_majflt,
                             vol_ctx_switches=ru1.ru_nvcsw - ru0.ru_nvcsw, invol_ctx_switches=ru1.ru_nivcsw - ru0.ru_nivcsw)
    out["side_stream_ratio"] = round(float(getattr(model, "side_stream_ratio", 0.0)), 3)   # < 1.4: the photo cell's stream runs beside the main one
    out.update(process_group=pg, rank_seconds=[round(v, 4) for v in rank_elapsed])
    # compact forms, read by main() into the line's LAST object (`evidence`): every bracketed kernel's time per step and every
    # roofline as {frac, bound, avg_launch_ms, bytes | flops per launch, counter traffic}
    out["kernel_ms_per_step"] = kms
    out["rooflines"] = {k: dict(frac=roofs[k]["frac"], bound=roofs[k]["bound"], avg_launch_ms=roofs[k]["avg_launch_ms"],
                                ms_per_step=roofs[k]["ms_per_step"],
                                per_launch=round(roofs[k].get("algorithmic_bytes_per_call", roofs[k].get("algorithmic_flops_per_call", 0))
                                                 * calls / max(1, roofs[k]["launches"])),
                                traffic=round(roofs[k]["traffic"]) if roofs[k].get("traffic") else None, **({"mfma_frac": roofs[k]["mfma_frac"]} if "mfma_frac" in roofs[k] else {}))
                        for k in order}
    trainer.release_host()
    del trainer, model, L
    gc.collect()
    torch.cuda.empty_cache()
    return out, kw


ALSO_CASES = [   # (name, command-line overrides, steps, warm-up: ~0.1-0.15 s of warm-up each, see --warmup)
    ("configs1_forward_fp32", ["--forward-only", "--precision", "f32"], 5, 6),
    ("configs1_forward_bf16", ["--forward-only"], 20, 40),
    ("configs1_forward_bf16x3", ["--forward-only", "--precision", "bf16x3"], 8, 8),
    ("train_bf16x3", ["--precision", "bf16x3"], 4, 4),
    ("train_ragged_lengths", ["--variant", "ragged"], 20, 30),
    ("configs4_long_album", ["--config", "long_album"], 3, 2),
    ("token_id_entry", ["--front-end"], 10, 10),
    ("time_warp_5", ["--time-warp", "5"], 10, 10),
    ("published_flag_set", ["--front-end", "--char-emb-size", "100", "--time-warp", "5"], 6, 6),
    ("headline_gc_default", ["--gc", "default"], 20, 10),       # the headline with CPython's collector left alone (Trainer gc_freeze off)
]


def also_helper():
    """The `also` orchestrator: started by the headline process BEFORE that process touches the GPU (a process that has
    initialised the GPU may not start other programs on these boxes), it waits for "go" on stdin, then runs every other
    configuration as its own `python bench.py ... --also off` child -- a fresh process each, so that no case inherits the
    stream-to-hardware-queue mapping, allocator state or clocks of the one before (in one process the ragged case ran at
    8.8 ms after three other models against 4.9-5.0 ms alone) -- and prints ONE JSON object with their compact results."""
    if sys.stdin.readline().strip() != "go":
        return
    only = [v for v in os.environ.get("FVTA_BENCH_ALSO", "").split(",") if v]      # (diagnostics: a subset of the cases)
    also = {}
    for name, over, steps, warm in ALSO_CASES:
        if only and name not in only:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warm),
               "--also", "off", "--no-cpu-baseline", "--gc", os.environ.get("FVTA_BENCH_GC", "freeze")] + over
        t0 = time.perf_counter()
        try:
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=900,
                                env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
            line = [l for l in pr.stdout.splitlines() if l.startswith("{")]
            if pr.returncode or not line:
                raise RuntimeError("exit code %d, %d JSON lines" % (pr.returncode, len(line)))
            r = json.loads(line[-1])
            also[name] = dict(value=r["value"], ms_per_step=r["ms_per_step"], event_median=r["ms_per_step_event_median"],
                              steps=steps, warmup=warm, dtype=r["dtype"], args=" ".join(over),
                              kernel_ms={k.replace("lstm_step_", "").replace("_main", ""): round(v, 3)
                                         for k, v in r["kernel_ms_per_step"].items() if v},
                              rooflines={k: [v["frac"], v["avg_launch_ms"]] for k, v in r["rooflines"].items()},
                              step_ms=[r["step_ms"][k] for k in ("min", "median", "max")],
                              host_ms=r["host_enqueue_ms_per_step"],
                              gc=[r["host_watch"]["gc_mode"], len(r["host_watch"]["gc_collections"]), r["host_watch"]["major_faults"]])
        except Exception as exc:   # a side measurement must not take the headline down
            also[name] = dict(error=repr(exc)[:200], args=" ".join(over))
    print(json.dumps(also), flush=True)


def main():
    args = parse()
    if args.also_helper:
        return also_helper()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # never returns
    default_headline = (args.config == "metric" and args.variant == "dense" and args.precision == "bf16" and not args.forward_only
                        and args.graph == "fvta" and not args.front_end and not args.batch and args.scaling == "weak"
                        and not args.time_warp)
    helper = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and (args.also == "on" or (args.also == "auto" and default_headline)):
        # the `also` orchestrator must exist before this process initialises the GPU; it sleeps until told to go
        helper = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--also-helper"], stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, text=True, env=dict(os.environ, FVTA_BENCH_GC=args.gc))
    # stdout carries ONE JSON line: whatever a C library prints there (RCCL's version banner, flushed at exit) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    from fvta_memexqa_amd import _lib, dist

    ws, rank, local = dist.init()
    if ws != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or plain `python bench.py --gpus N`, "
                         "which spawns them)" % (args.gpus, ws))
    local %= max(1, torch.cuda.device_count())      # (more ranks than GPUs only under FVTA_DIST_BACKEND=gloo, a test set-up)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.load()

    # achievable HBM read rate on THIS device: one coalesced read-only pass over 1.5 GiB, median of 5
    probe_gbs = None
    if rank == 0:
        try:
            pbuf = torch.empty(3 << 28, dtype=torch.float32, device=dev).zero_()       # 1.5 GiB, about the context tensor
            sink = torch.zeros(16, dtype=torch.float32, device=dev)
            strm = torch.cuda.current_stream().cuda_stream
            ts = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.fvta_probe_hbm_read(pbuf.data_ptr(), pbuf.numel() * 4, sink.data_ptr(), strm)
                e1.record()
                e1.synchronize()
                if i:
                    ts.append(e0.elapsed_time(e1))
            probe_gbs = round(pbuf.numel() * 4 / (statistics.median(ts) * 1e-3) / 1e9, 1)
            del pbuf
            torch.cuda.empty_cache()
        except Exception as exc:                                                        # measurement aid only
            log("hbm probe failed: %r" % (exc,))

    out, kw = run_case(args, lib, ws, rank, local, probe_gbs)
    if rank != 0:
        dist.shutdown()
        return
    if helper is not None:
        import gc
        gc.collect()
        torch.cuda.empty_cache()       # (this process keeps its context; the cases get the whole memory)
        try:
            helper.stdin.write("go\n")
            helper.stdin.flush()
            line = helper.stdout.readline()
            helper.wait(timeout=60)
            out["also"] = json.loads(line)
        except Exception as exc:
            out["also"] = dict(error=repr(exc))
        out["also_note"] = "each case: a fresh `python bench.py <args> --also off` process right after the headline; rooflines = [frac of HBM (lstm_dw: of MFMA), avg_launch_ms]; step_ms = [min, median, max]"
    if ws == 1 and not args.no_cpu_baseline and args.graph == "fvta":
        log('timing the CPU oracle (%d threads of %d host CPUs)' % (args.cpu_threads, host_cores()))
        out["cpu_baseline"] = cpu_baseline(kw, args.cpu_sample, args.forward_only, args.cpu_threads)
    else:
        out["cpu_baseline"] = None
    # LAST in the line (the driver's record keeps the line's tail): the numbers a reader needs, compact
    roofs = out.pop("rooflines", {})
    kms = out.pop("kernel_ms_per_step", {})
    if args.also == "off":           # a child of the `also` orchestrator: its parent reads these two
        out["rooflines"], out["kernel_ms_per_step"] = roofs, kms
    ev = dict(headline=dict(ms_per_step=out["ms_per_step"], event_median=out["ms_per_step_event_median"], value=out["value"],
                            kernel_ms_per_step=kms, rooflines=roofs))
    for k in ("also", "also_note"):          # (the cases sit right in front of `evidence`, behind the CPU baseline)
        if k in out:
            out[k] = out.pop(k)
    out["evidence"] = ev
    json_out.write(json.dumps(out, separators=(",", ":")) + "\n")
    json_out.flush()
    dist.shutdown()


if __name__ == "__main__":
    main()
