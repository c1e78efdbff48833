"""Reader (and a minimal writer) for TensorFlow's V2 checkpoint files -- what the reference's `saver.save` / `saver.restore`
(main.py:296, 430-440, 640-665) exchange: `<prefix>.index` + `<prefix>.data-00000-of-00001`, and the `checkpoint` state file
that names the latest prefix.

PARITY UNPINNED: there is no TensorFlow in this image and the reference ships no checkpoint, so the format below is a
restatement of TensorFlow's published on-disk layout (tensorflow/core/lib/io/{table,block,format}.cc -- the LevelDB
table format -- and tensorflow/core/util/tensor_bundle + protobuf/tensor_bundle.proto), checked only by a
writer -> reader round trip (tests/test_host_logic.py).  Layout:

  .index   a LevelDB-style sorted table, uncompressed blocks:
             block   = entries, restart offsets (uint32 each), restart count (uint32), then a 5-byte trailer
                       (compression type 0, masked crc32c of block + type)
             entry   = varint shared-key-bytes, varint unshared-key-bytes, varint value-bytes, key delta, value
             footer  = 48 bytes: metaindex handle, index handle (varint offset, varint size each), zero padding to 40,
                       magic 0xdb4775248b80fb57 (little endian)
           key ""            -> BundleHeaderProto  {1: num_shards, 2: endianness, 3: version}
           key <tensor name> -> BundleEntryProto   {1: dtype, 2: shape {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size,
                                                    6: crc32c (fixed32, masked)}
  .data-00000-of-00001   the tensors' raw little-endian bytes at [offset, offset + size).
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_, 4: np.uint8, 6: np.int8, 5: np.int16}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---------------------------------------------------------------- varints / protobuf wire format
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """yield (field number, wire type, value) of a serialized message: varint -> int, length-delimited -> bytes,
    fixed32 / fixed64 -> int"""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v, pos = bytes(buf[pos:pos + n]), pos + n
        elif wt == 5:
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        elif wt == 1:
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for f, _, v in _proto_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:                                   # TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = _signed(v3)
                    e["shape"].append(size)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:                                   # repeated TensorSliceProto: a partitioned variable
            e["slices"] += 1
    return e


# ---------------------------------------------------------------- crc32c (Castagnoli), LevelDB masking
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    return _CRC_TABLE


def _crc_raw(data, c):
    """the CRC register after `data`, starting from register c (no pre / post inversion): byte at a time"""
    tbl = _crc_table()
    for b in bytes(data):
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


def _gf2_apply(cols, v):
    """cols[i] = image of bit i; the image of v"""
    r, i = 0, 0
    while v:
        if v & 1:
            r ^= cols[i]
        v >>= 1
        i += 1
    return r


_SHIFT_CACHE = {}


def _shift_tables(L):
    """the operator "advance the crc register over L zero bytes" as four 256-entry tables (one zero byte, then
    square-and-multiply on L); cached per chunk length -- a checkpoint's tensors repeat a handful of sizes"""
    t = _SHIFT_CACHE.get(L)
    if t is None:
        one = [_crc_raw(b"\0", 1 << i) for i in range(32)]
        op, sq, e = [1 << i for i in range(32)], one, L
        while e:
            if e & 1:
                op = [_gf2_apply(sq, col) for col in op]
            sq = [_gf2_apply(sq, col) for col in sq]
            e >>= 1
        t = [[_gf2_apply(op, b << (8 * k)) for b in range(256)] for k in range(4)]
        if len(_SHIFT_CACHE) < 64:
            _SHIFT_CACHE[L] = t
    return t


def _native_crc32c():
    """a C implementation when one is importable (crc32c / google_crc32c: neither is in this image): (data, crc) -> crc"""
    try:
        import crc32c as _m
        return lambda data, crc: _m.crc32c(data, crc)
    except Exception:
        pass
    try:
        import google_crc32c as _g
        return lambda data, crc: _g.extend(crc, bytes(data))
    except Exception:
        return None


_NATIVE = _native_crc32c()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of `data`, continuing from `crc`.  Long buffers (a checkpoint's LSTM kernels and optimiser
    slots are tens of MB) are cut into 4096 equal chunks whose registers advance TOGETHER, one numpy table lookup per
    byte position, and are then folded left to right: the register is linear over GF(2), so
    reg(A || B) = shift_len(B)(reg(A)) xor reg_0(B), with the shift by a chunk's length as four 256-entry tables."""
    import numpy as np
    if isinstance(data, np.ndarray):
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)      # (no copy when it is contiguous already)
    else:
        buf = np.frombuffer(data, dtype=np.uint8)                        # bytes / bytearray / memoryview: no copy
    if _NATIVE is not None:
        return _NATIVE(buf, crc)
    n, lanes = buf.size, 4096
    c = crc ^ 0xFFFFFFFF
    if n < 64 * lanes:
        return _crc_raw(buf.tobytes(), c) ^ 0xFFFFFFFF
    L = n // lanes
    tbl = np.array(_crc_table(), dtype=np.uint32)
    # byte position i of every lane as ONE contiguous row (a column of the [lanes, L] view is a 4096-element strided gather
    # per step: the transposed copy is a single pass)
    body = np.ascontiguousarray(buf[:L * lanes].reshape(lanes, L).T)
    reg = np.zeros(lanes, dtype=np.uint32)
    for i in range(L):                      # every lane's register from 0 over its own chunk
        reg = tbl[(reg ^ body[i]) & 0xFF] ^ (reg >> np.uint32(8))
    shift = _shift_tables(L)
    for r in reg.tolist():
        c = shift[0][c & 0xFF] ^ shift[1][(c >> 8) & 0xFF] ^ shift[2][(c >> 16) & 0xFF] ^ shift[3][c >> 24] ^ r
    return _crc_raw(buf[L * lanes:].tobytes(), c) ^ 0xFFFFFFFF


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- table reader
def _read_block(f, offset, size):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) < size + 5:
        raise ValueError("truncated table block")
    if raw[size] != 0:
        raise ValueError("compressed table block (type %d): TensorFlow writes checkpoints uncompressed" % raw[size])
    if struct.unpack_from("<I", raw, size + 1)[0] != _mask(crc32c(raw[:size + 1])):    # block + type byte, masked
        raise ValueError("table block at offset %d fails its crc32c" % offset)
    return raw[:size]


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _handle(buf, pos=0):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_index(index_path):
    """-> (header dict, {tensor name: entry dict})"""
    with open(index_path, "rb") as f:
        f.seek(0, os.SEEK_END)
        total = f.tell()
        if total < 48:
            raise ValueError("%s is too short for a table" % index_path)
        f.seek(total - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != MAGIC:
            raise ValueError("%s is not a TensorFlow V2 checkpoint index (bad table magic)" % index_path)
        _, _, pos = _handle(footer)                    # metaindex (empty)
        ioff, isize, _ = _handle(footer, pos)
        header, entries = {}, {}
        for _, hv in _block_entries(_read_block(f, ioff, isize)):
            boff, bsize, _ = _handle(hv)
            for key, val in _block_entries(_read_block(f, boff, bsize)):
                if key == b"":
                    header = {fld: v for fld, _, v in _proto_fields(val)}
                else:
                    entries[key.decode()] = _parse_entry(val)
    return header, entries


def latest_checkpoint(path):
    """tf.train.get_checkpoint_state(dir).model_checkpoint_path (main.py:641-645): the prefix the `checkpoint` state
    file names; a prefix (or its .index file) passes through."""
    if os.path.isdir(path):
        state = os.path.join(path, "checkpoint")
        if not os.path.exists(state):
            raise Exception("Model not exists")                                  # main.py:665
        for line in open(state):
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                return p if os.path.isabs(p) else os.path.join(path, p)
        raise Exception("Model not exists")
    return path[:-6] if path.endswith(".index") else path


def read_checkpoint(path, names=None, verify=True):
    """{variable name: ndarray} of a V2 checkpoint (`path`: directory with a `checkpoint` state file, a prefix, or the
    .index file).  `names`: read only these.  `verify`: check every tensor's masked crc32c (the index blocks' are always
    checked); entries that carry slices (partitioned variables: the reference saves none) are rejected."""
    prefix = latest_checkpoint(path)
    if not os.path.exists(prefix + ".index"):
        raise Exception("Model not exists")                                      # main.py:665
    header, entries = read_index(prefix + ".index")
    nshards = max(1, header.get(1, 1))
    out, files = {}, {}
    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e["dtype"] not in _DTYPES:
                continue                                # strings etc.: nothing the model stores
            if e["slices"]:
                raise ValueError("%s is stored in slices (a partitioned variable): not supported" % name)
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, nshards), "rb")
            f = files[sid]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError("truncated tensor data for %s" % name)
            if verify and e["crc32c"] is not None and e["crc32c"] != _mask(crc32c(raw)):
                raise ValueError("tensor data of %s fails its crc32c" % name)
            out[name] = np.frombuffer(raw, dtype=np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


# ---------------------------------------------------------------- writer (tests / export)
def _field(num, wt, payload):
    return _put_varint((num << 3) | wt) + payload


def _entry_proto(arr, offset, crc):
    shape = b"".join(_field(2, 2, (lambda d: _put_varint(len(d)) + d)(_field(1, 0, _put_varint(int(s))))) for s in arr.shape)
    msg = _field(1, 0, _put_varint(_DTYPE_IDS[arr.dtype]))
    msg += _field(2, 2, _put_varint(len(shape)) + shape)
    if offset:
        msg += _field(4, 0, _put_varint(offset))
    msg += _field(5, 0, _put_varint(arr.nbytes))
    msg += _field(6, 5, struct.pack("<I", crc))
    return msg


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (key, val) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_checkpoint(prefix, tensors, block_bytes=4096, state_file=True):
    """Write {name: ndarray} as a single-shard V2 checkpoint at `prefix` (+ the `checkpoint` state file beside it)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = []
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        off = 0
        for name in sorted(tensors):
            a = np.ascontiguousarray(tensors[name])
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            raw = a.tobytes()
            f.write(raw)
            items.append((name.encode(), _entry_proto(np.asarray(tensors[name]), off, _mask(crc32c(raw)))))
            off += len(raw)
    header = _field(1, 0, _put_varint(1)) + _field(3, 2, (lambda v: _put_varint(len(v)) + v)(_field(1, 0, _put_varint(1))))
    items = [(b"", header)] + items                       # "" sorts first
    with open(prefix + ".index", "wb") as f:
        index_items, cur, pos = [], [], 0

        def flush():
            nonlocal cur, pos
            if not cur:
                return
            blk = _build_block(cur)
            f.write(blk + b"\x00" + struct.pack("<I", _mask(crc32c(blk + b"\x00"))))
            index_items.append((cur[-1][0], _put_varint(pos) + _put_varint(len(blk))))
            pos += len(blk) + 5
            cur = []
        size = 0
        for it in items:
            cur.append(it)
            size += len(it[0]) + len(it[1]) + 3
            if size >= block_bytes:
                flush()
                size = 0
        flush()
        meta = _build_block([])
        meta_off = pos
        f.write(meta + b"\x00" + struct.pack("<I", _mask(crc32c(meta + b"\x00"))))
        pos += len(meta) + 5
        idx = _build_block(index_items, restart_interval=1)
        idx_off = pos
        f.write(idx + b"\x00" + struct.pack("<I", _mask(crc32c(idx + b"\x00"))))
        footer = _put_varint(meta_off) + _put_varint(len(meta)) + _put_varint(idx_off) + _put_varint(len(idx))
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    if state_file:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix
