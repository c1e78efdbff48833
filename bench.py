#!/usr/bin/env python3
"""Benchmark of the FVTA hot path on MI355X: QA-pairs/s for one training step
(forward + backward + optimiser update) at BASELINE.json's metric shape
(batch 64 per GPU, 40 photos x 5 text streams x 30 tokens, hidden 512).

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU; albums/questions are sharded over ranks (weak scaling: 64 QA
pairs per GPU), the only collective is the all-reduce of the flat gradient
buffer.  Inputs (synthetic, seeded) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the bi-LSTM
step GEMM, MFMA bound); `roofline_attention` is the fused focal-attention
kernel against HBM.  Kernel durations come from HIP events recorded on the
launch stream inside the library (fvta_profile_*), live over the timed region.
`cpu_baseline` times the CPU oracle ("port") on a bounded sample of the same
workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3        # fp32 MFMA (v_mfma_f32_32x32x2_f32)
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA


def log(msg):
    if os.environ.get("FVTA_BENCH_VERBOSE", "1") != "0":
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
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
    return ap.parse_args()


def lstm_flops(spec, B, J, din, d):
    """algorithmic flops of one bi-LSTM forward call, dense: 2 dirs x steps x 2*B*(in+d)*4d (h part absent at t=0)"""
    return 2 * (J * 2.0 * B * (din + d) * 4 * d - 2.0 * B * d * 4 * d)


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


def main():
    args = parse()
    from fvta_memexqa_amd import _lib, dist
    from fvta_memexqa_amd.model_v2 import Model
    from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs
    from fvta_memexqa_amd.trainer import Trainer

    ws, rank, local = dist.init()
    if ws != args.gpus and ws > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, ws))
    local %= max(1, torch.cuda.device_count())      # (more ranks than GPUs only under FVTA_DIST_BACKEND=gloo, a test set-up)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.load()

    kw = dict(CONFIGS[args.config], dense=(args.variant == "dense"))
    if args.batch:
        kw["N"] = args.batch
    if args.scaling == "strong":      # fixed total work: global/ws QA pairs per rank
        lo, hi = dist.shard_range(args.global_batch, ws, rank)
        kw["N"] = hi - lo
    spec = SynthSpec(**kw)
    cfg = dict(spec.cfg(), batch_size=spec.N, precision=args.precision, optimizer=args.optimizer,
               init_lr=0.001 if args.optimizer == "adam" else 0.5)
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

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log('warm-up step %d done' % i)
    # ---- the timed region: exactly `steps` steps between barrier + synchronize pairs, no profiling brackets inside;
    # one HIP event per step on the main stream (every side stream has joined it when a step ends) gives the
    # per-step device times whose median is reported beside the wall-clock mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_elapsed = dist.gather_over_ranks(elapsed, dev)      # every rank's own wall clock over the timed region
    elapsed = dist.max_over_ranks(elapsed, dev)
    pg = dist.describe()
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    log('timed region done: %.3f s for %d steps (per-step HIP events: median %.3f ms)'
        % (elapsed, args.steps, statistics.median(step_ms)))
    # ---- a second pass of the same steps with the library's HIP-event brackets around the kernels the rooflines
    # are quoted for (events on the launch streams); kept out of the headline
    lib.fvta_profile_enable(1)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    lib.fvta_profile_enable(0)
    dist.barrier()

    def collect(pid):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        lib.fvta_profile_collect(pid, ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value

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
        except Exception as exc:                                                        # measurement aid only
            log("hbm probe failed: %r" % (exc,))

    prof = {name: collect(pid) for name, pid in [("lstm_step_fwd", 1), ("lstm_step_bwd", 2), ("lstm_dw", 3),
                                                 ("attn_fwd_main", 4), ("attn_bwd_main", 5), ("lstm_dx", 6),
                                                 ("lstm_step_fwd_photo_cell", 17)]}
    if rank != 0:
        dist.shutdown()
        return
    total_qa = spec.N * ws * args.steps
    value = total_qa / elapsed
    # ---- roofline of the dominant kernel family: the bi-LSTM step kernels of the text cell.
    # lstm_step_fwd (GEMM + fused gate epilogue) per launch moves, per active (row, direction):
    #   x shadow in_i*2 B + h shadow read d*2 + c_prev d*4 + c d*4 + h (fp32, into the context tensor) d*4
    #   + h shadow d*2 + saved gates 4d*2      
    # and does 2*(in+d)*4d flops.  Intensity 265 flop/B < the machine's 312 (2.5 PFLOP/s / 8 TB/s): HBM bound.
    T = L.groups["text"]
    calls = args.steps
    lens = T.lens.float()
    dp = model.dp
    in_i = ((spec.text_in + 1 + 31) // 32) * 32
    fl_text = float((2 * (lens * 2.0 * (spec.text_in + dp) * 4 * dp - (lens > 0).float() * 2.0 * dp * 4 * dp)).sum().item())
    by_text = float((2 * lens * (in_i * 2 + dp * 2 + dp * 4 * 3 + dp * 2 + 4 * dp * 2)).sum().item())
    ms_f, n_f = prof["lstm_step_fwd"]
    peak_tf = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    roof = None
    if n_f:
        tf = fl_text * calls / (ms_f * 1e-3) / 1e12
        gbs = by_text * calls / (ms_f * 1e-3) / 1e9
        wreg = args.precision == "bf16" and dp % 128 == 0 and (dp, in_i) in ((512, 224), (512, 128), (1024, 224), (1024, 128), (128, 128), (128, 32))
        roof = dict(kernel=("lstm_fwd_wreg_bf16 (forward step, weights in registers)" if wreg else "lstm_step_fwd_%s" % args.precision), bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS,
                    unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None, algorithmic_bytes_per_call=by_text,
                    mfma_tflops=round(tf, 1), mfma_frac=round(tf / peak_tf, 4), launches=n_f,
                    avg_launch_ms=round(ms_f / n_f, 4))
        # rocprofv3 --stats averages per kernel SYMBOL: the photo cell (side stream, small launches) runs the same
        # lstm_step_fwd symbol, so its launches are reported too -- the all-launch mean is the figure to hold against
        # the profiler's AverageNs (profiles/README.md)
        ms_p, n_p = prof["lstm_step_fwd_photo_cell"]
        if n_p:
            roof.update(launches_photo_cell=n_p, avg_launch_ms_photo_cell=round(ms_p / n_p, 4),
                        avg_launch_ms_all_launches=round((ms_f + ms_p) / (n_f + n_p), 4))
        if args.precision == "f32":   # exact-fp32 MFMA: 64 flop/clk/SIMD -- the matrix pipe bounds this engine, not HBM
            roof.update(bound="mfma", achieved=round(tf, 1), peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=round(tf / peak_tf, 4),
                        hbm_gbs=round(gbs, 1))
    # ---- attention kernel against HBM: algorithmic bytes = valid rows * w * 4 + question + output
    valid_rows = int(L.hall_mask.sum().item())
    att_bytes = (valid_rows * model.wp + spec.N * L.JQ * model.wp + spec.N * model.wp) * 4.0
    ms_a, n_a = prof["attn_fwd_main"]
    roof_att = None
    if n_a and args.graph == "fvta":   # (model.py's graph runs seven large 1-D attentions under the same bracket)
        # (the library's profile bracket covers the context attention only, not the K = 1 question attention)
        per_step_ms = ms_a / args.steps
        gbs = att_bytes / (per_step_ms * 1e-3) / 1e9
        roof_att = dict(kernel=(("attn_fwd_pair16" if model.wp >= 512 and model.simi != 4 and os.environ.get("FVTA_ATTN_WAVE16", "3") in ("2", "3")
                                 else "attn_fwd_rows16") if (L.JQ <= 32 and 128 <= model.wp <= 1024) else "attn_fwd_main")
                        + " (fvta_attn_fwd main kernel)", bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None, algorithmic_bytes=att_bytes,
                        ms_per_step=round(per_step_ms, 4))
    # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
    # tools/bench_lstm.py / tools/bench_attn.py at this same shape; FETCH_SIZE doubled per the gfx950 note of
    # MI355X_MICROARCH.md).  Not collected live: counters need their own profiler pass.
    def pmc_traffic(fname, key):
        try:
            d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", fname)))
            for k, v in d.items():
                if key in k:
                    return v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
        except Exception:
            pass
        return None
    for r_ in (roof, roof_att):
        if r_ is not None and probe_gbs and r_["bound"] == "hbm":
            r_["achievable_peak"] = probe_gbs
            r_["frac_of_achievable"] = round(r_["achieved"] / probe_gbs, 4)
    dense_metric = args.config == "metric" and args.variant == "dense" and not args.batch and args.graph == "fvta"
    if roof is not None and dense_metric and args.precision == "bf16":
        roof["traffic"] = pmc_traffic("r03_lstm_pmc.json", "lstm_fwd_wreg_bf16<fvta::WregCfg<14, 32, 2>")  # the text cell's instantiation
        roof["traffic_note"] = "bytes per launch, profiles/r03_lstm_pmc.json; algorithmic per launch = algorithmic_bytes_per_call / 30"
    # the fused backward step (now the longest kernel family of the step): per active (row, unit) it reads the saved bf16
    # gates (8 B), c_{t-1} (4), d_out (4) and the running dc (4) and writes dz (8) and dc (4): 32 B -- DESIGN.md 4.3
    roof_bwd = None
    ms_b, n_b = prof["lstm_step_bwd"]
    if n_b and args.precision == "bf16" and not args.forward_only:
        by_bwd = float((2 * lens * dp * 32).sum().item())
        gbs_b = by_bwd * calls / (ms_b * 1e-3) / 1e9
        roof_bwd = dict(kernel="lstm_bwd_fused_bf16 (backward step of the text cell: dz(t+1) Wh^T + gate gradient)", bound="hbm",
                        achieved=round(gbs_b, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs_b / PEAK_HBM_GBS, 4), traffic=None,
                        algorithmic_bytes_per_call=by_bwd, ms_per_step=round(ms_b / args.steps, 4), launches=int(n_b),
                        avg_launch_ms=round(ms_b / max(1, n_b), 4),
                        note="algorithmic bytes: the gate-gradient epilogue (32 B per row and unit: gates 8, c 4, d_out 4, dc 4 + 4, "
                             "dz 8); the k-loop's re-read of dz(t+1) (8 B) not counted; the dx pass has its own bracket (lstm_dx)")
    if roof_bwd is not None and dense_metric:
        roof_bwd["traffic"] = pmc_traffic("r03_lstm_bwd_pmc.json", "lstm_bwd_fused_bf16<2, 4, 1>")
        roof_bwd["traffic_note"] = "bytes per launch, profiles/r03_lstm_bwd_pmc.json; algorithmic per launch = algorithmic_bytes_per_call / 30"
    if roof_att is not None and dense_metric:
        roof_att["traffic"] = pmc_traffic("r02c_attention_pmc.json", "attn_fwd_pair16")
        roof_att["traffic_note"] = "bytes per launch, profiles/r02c_attention_pmc.json"
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
                    + (" -- model.py's soft-attention baseline graph (multi-layer + multi-modal + direct-link + choices + "
                       "question attention) instead of the FVTA model" if args.graph == "model_py" else ""),
                    qa_pairs_per_gpu=spec.N, albums=spec.A, photos=spec.P, text_streams=spec.S, tokens=spec.L, hidden=spec.d,
                    global_batch=spec.N * ws, K=L.K, T=L.T, JQ=L.JQ,
                    parallelism="dp%d (QA pairs sharded, flat-gradient all-reduce)" % ws),
        roofline=roof, roofline_attention=roof_att, roofline_lstm_bwd=roof_bwd,
        process_group=pg, rank_seconds=[round(v, 4) for v in rank_elapsed],
        kernel_ms_per_step={k: round(v[0] / args.steps, 4) for k, v in prof.items()},
        kernel_ms_note="HIP-event brackets on the launch streams, over a second pass of the same %d steps right after the "
                       "timed region (the timed region itself carries no brackets)" % args.steps,
    )
    if ws == 1 and not args.no_cpu_baseline and args.graph == "fvta":
        log('timing the CPU oracle (%d threads of %d host CPUs)' % (args.cpu_threads, host_cores()))
        out["cpu_baseline"] = cpu_baseline(kw, args.cpu_sample, args.forward_only, args.cpu_threads)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    dist.shutdown()


if __name__ == "__main__":
    main()
