"""Host-side logic that needs no GPU: synthetic spec shapes, hidden-size padding maps, parameter
naming, product/oracle separation."""
import os
import re

import numpy as np
import pytest
import torch

from fvta_memexqa_amd.synth import CONFIGS, SynthSpec, make_inputs, make_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_shapes_match_survey_table():
    m = SynthSpec(**CONFIGS["metric"])
    assert (m.K, m.JMAX, m.T, m.w) == (6, 1200, 1200, 1024)          # hall [64,6,1,1200,1024]
    p = SynthSpec(**CONFIGS["plumbing"])
    assert (p.N, p.K, p.A, p.JMAX, p.w) == (4, 3, 2, 50, 256)        # hall [4,3,2,50,256]
    s = SynthSpec(**CONFIGS["long_album"])
    assert (s.K, s.T, s.w) == (7, 7200, 2048)                        # hall [32,7,1,7200,2048]


def test_synth_inputs_are_seeded_and_ragged():
    spec = SynthSpec(N=3, A=2, P=4, S=2, L=6, d=8, dense=False, text_in=8, img_in=4)
    a, b = make_inputs(spec), make_inputs(spec)
    assert torch.equal(a["ctx"][0]["x"], b["ctx"][0]["x"])
    c = make_inputs(spec, rank=1)
    assert not torch.equal(a["ctx"][0]["x"], c["ctx"][0]["x"])
    m = a["ctx"][0]["mask"]
    ln = m.sum(-1)
    assert (m == (torch.arange(6) < ln[..., None])).all()            # prefix masks (model_v2.py:1346-1358)
    assert a["y"].sum(1).eq(1).all()
    dense = make_inputs(SynthSpec(N=2, A=1, P=3, S=1, L=4, d=8, dense=True, text_in=8, img_in=4))
    assert dense["ctx"][0]["mask"].all() and dense["q"]["mask"].all()


def test_param_shapes_follow_reference():
    spec = SynthSpec(N=2, A=1, P=3, S=1, L=4, d=16, text_in=12, img_in=8, simiMatrix=2)
    p = make_params(spec)
    assert p["text_kernel"].shape == (12 + 16, 64) and p["image_kernel"].shape == (8 + 16, 64)
    assert p["att_W"].shape == (2 * 32, 1) and p["out_W"].shape == (5 * 32, 1)
    assert float(p["att_W"].abs().max()) <= 0.2 + 1e-6               # truncated normal, 2 sigma (model_v2.py:88)
    assert float(p["text_bias"].abs().max()) == 0.0


def test_padded_hidden_map():
    from fvta_memexqa_amd.model_v2 import padded_hidden
    assert [padded_hidden(d) for d in (20, 32, 50, 100, 128, 512, 1000)] == [32, 32, 64, 128, 128, 512, 1024]
    with pytest.raises(ValueError):
        padded_hidden(2000)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
    leg may touch it."""
    pkg = os.path.join(ROOT, "fvta_memexqa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle", bench)]
    # the fused and the literal oracle, both imported inside cpu_baseline() and nowhere else in bench.py
    lo, hi = bench.index("def cpu_baseline"), bench.index("def main")
    assert len(uses) == 2 and all(lo < u < hi for u in uses)


def test_no_reference_sources_in_repo():
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__")]
        for f in files:
            if f.endswith(".py") and f != os.path.basename(__file__):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*import tensorflow", src, flags=re.M), os.path.join(dirpath, f)


def test_tf_checkpoint_reader_round_trip(tmp_path):
    """tf_checkpoint.py: the V2 checkpoint layout (LevelDB-style index table + raw data shard) written and read back --
    many variables (several index blocks, shared key prefixes), scalars, int64; the `checkpoint` state file resolves the
    prefix like tf.train.get_checkpoint_state (main.py:641-645).  Format restated from TensorFlow's sources: unpinned."""
    from fvta_memexqa_amd import tf_checkpoint as tc
    assert tc.crc32c(b"123456789") == 0xE3069283                     # the CRC-32C check value
    rng = np.random.RandomState(0)
    tensors = {"model_x/global_step": np.int64(1234)}
    for i in range(70):
        tensors["model_x/reader/text/utext/fw/cell_%02d/kernel" % i] = rng.randn(3 + i % 5, 4).astype(np.float32)
        tensors["model_x/reader/text/utext/fw/cell_%02d/kernel/Adadelta" % i] = rng.randn(3 + i % 5, 4).astype(np.float32)
    tensors["model_x/output/choicelogits/b"] = np.zeros(1, np.float32)
    prefix = tc.write_checkpoint(str(tmp_path / "model-1234"), tensors, block_bytes=512)
    header, entries = tc.read_index(prefix + ".index")
    assert header[1] == 1 and len(entries) == len(tensors)
    for src in (str(tmp_path), prefix, prefix + ".index"):
        got = tc.read_checkpoint(src)
        assert set(got) == set(tensors)
        for k, v in tensors.items():
            assert got[k].dtype == np.asarray(v).dtype and np.array_equal(got[k], v), k
    assert int(got["model_x/global_step"]) == 1234
    only = tc.read_checkpoint(prefix, names={"model_x/output/choicelogits/b"})
    assert list(only) == ["model_x/output/choicelogits/b"]
    os.makedirs(str(tmp_path / "empty"))
    for missing in (str(tmp_path / "empty"), str(tmp_path / "model-999")):
        with pytest.raises(Exception, match="Model not exists"):
            tc.read_checkpoint(missing)


def test_tf_checkpoint_reader_rejects_corruption(tmp_path):
    """Every table block and every tensor carries a masked crc32c (LevelDB block trailer; BundleEntryProto field 6): a
    flipped byte in the data shard or in the index is an error, not a silently different weight; entries stored in
    slices (field 7) are refused."""
    from fvta_memexqa_amd import tf_checkpoint as tc
    assert tc._mask(tc.crc32c(b"123456789")) == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    rng = np.random.RandomState(1)
    tensors = {"m/a": rng.randn(5, 3).astype(np.float32), "m/b": rng.randn(7).astype(np.float32)}
    prefix = tc.write_checkpoint(str(tmp_path / "model-1"), tensors)
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[9] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        tc.read_checkpoint(prefix)
    got = tc.read_checkpoint(prefix, verify=False)                    # the caller may opt out (large files: pure-Python CRC)
    assert not np.array_equal(got["m/a"], tensors["m/a"]) or not np.array_equal(got["m/b"], tensors["m/b"])
    prefix = tc.write_checkpoint(str(tmp_path / "model-2"), tensors)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[12] ^= 0x01                                                   # inside the first data block of the table
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="crc32c"):
        tc.read_checkpoint(prefix)
    e = tc._parse_entry(tc._entry_proto(tensors["m/a"], 0, 0) + tc._field(7, 2, tc._put_varint(0)))
    assert e["slices"] == 1


def test_bench_gpus_n_without_launcher_never_reports_one_rank():
    """`python bench.py --gpus 2` started plainly must either run two ranks (it spawns them itself, before the parent
    touches the GPU) or fail: a line that says n_gpus 1 for a --gpus 2 request would let a scaling record measure one GPU
    seven times.  Here (no GPU) the spawned ranks fail -> non-zero exit, nothing on stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert all(l.get("n_gpus") == 2 for l in lines)
    assert lines or r.returncode != 0
    # a launcher's world size that contradicts --gpus is refused too
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                        env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and not [l for l in r2.stdout.splitlines() if l.startswith("{")]


def test_crc32c_chunked_form_equals_bytewise():
    """tf_checkpoint.crc32c: the 4096-lane numpy form (long buffers) against the byte-at-a-time register, at the lane
    boundary sizes, with continuation from a previous crc; the Castagnoli check value."""
    from fvta_memexqa_amd import tf_checkpoint as tc
    assert tc.crc32c(b"123456789") == 0xE3069283
    rng = np.random.default_rng(0)
    for n in (0, 1, 100, 64 * 4096 - 1, 64 * 4096, 64 * 4096 + 5, 1_000_003):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        ref = tc._crc_raw(d, 0xFFFFFFFF) ^ 0xFFFFFFFF
        assert tc.crc32c(d) == ref, n
        assert tc.crc32c(d[n // 3:], tc.crc32c(d[:n // 3])) == ref, n


def test_get_batches_refuses_a_tail_smaller_than_the_world_before_the_first_batch():
    """data parallelism: num_examples % (batch_size * world) in (0, world) would leave a rank without a batch in the last
    group of an epoch -- refused at the first next(), not after an epoch of training"""
    from fvta_memexqa_amd.utils import Dataset
    data = {"q": list(range(9)), "idxs": list(range(9))}
    ds = Dataset.__new__(Dataset)
    ds.data, ds.datatype, ds.shared = data, "train", {}
    ds.valid_idxs, ds.num_examples = list(range(9)), 9
    with pytest.raises(ValueError):
        next(ds.get_batches(2, 10, rank=0, world=4, seed=3))      # 9 % 8 = 1 < 4


def test_bench_also_helper_exits_quietly_without_go():
    """bench.py's `also` orchestrator (started before the headline process touches the GPU) waits for "go" on stdin; when
    the headline process dies first it sees EOF and must leave without starting anything or printing a line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--also-helper"], input="", capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == ""


def test_trainer_owns_and_releases_the_collector():
    """Trainer.own_host (called by the first step_device under config gc_freeze, default on): CPython's collector frozen and
    off while a trainer steps, back on when it releases the host or goes away; gc_freeze=False never touches it."""
    import gc
    from fvta_memexqa_amd.trainer import Trainer
    assert gc.isenabled()
    t = Trainer(object(), dict(init_lr=0.5))
    assert t.gc_freeze and gc.isenabled()
    t.own_host()
    assert not gc.isenabled() and gc.get_freeze_count() > 0
    t.own_host()                      # idempotent
    t.release_host()
    assert gc.isenabled() and gc.get_freeze_count() == 0
    t.own_host()
    del t                             # a trainer that goes away gives the collector back
    assert gc.isenabled()
    assert not Trainer(object(), dict(gc_freeze=False)).gc_freeze
