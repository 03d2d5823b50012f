"""Reader (and a minimal writer) for the TensorFlow checkpoint files the reference saves.

The reference stores G / H weights with `tf.train.Saver` (v1: /root/reference/main_density_est.py:123,
389-411) or `tf.train.Checkpoint` + `CheckpointManager` (v2: /root/reference/src/pyroundtrip/roundtrip.py:
48-58).  Both write the TensorBundle format: `<prefix>.index`, an SSTable (LevelDB table format) mapping
tensor name -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}, and
`<prefix>.data-00000-of-00001` holding the raw little-endian tensor bytes.

No TensorFlow, no protobuf package: the SSTable blocks, the two small protos and (optional) snappy blocks
are decoded here by hand.  NOT VERIFIED against a TF-written file: none exists offline;
the writer below produces the same layout so that tests can round-trip and fixtures can be exchanged
with a TensorFlow install elsewhere.

Variable names:
  v1  <scope>/fully_connected/weights, <scope>/fully_connected/biases, <scope>/fully_connected_<i>/...
      for i = 1..nb_layers (tcl.fully_connected's default scopes under tf.variable_scope(name), model.py:63)
  v2  <attr>/all_layers/<i>/0/kernel/.ATTRIBUTES/VARIABLE_VALUE and .../bias/... (object-graph path of
      `self.all_layers.append([fc_layer, norm_layer])`, src/pyroundtrip/model.py:25) — expected, unverified.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64 = 1, 2, 3, 7, 9
_NP_OF_DT = {DT_FLOAT: np.dtype('<f4'), DT_DOUBLE: np.dtype('<f8'), DT_INT32: np.dtype('<i4'),
             DT_INT64: np.dtype('<i8')}
_DT_OF_NP = {np.dtype('float32'): DT_FLOAT, np.dtype('float64'): DT_DOUBLE, np.dtype('int32'): DT_INT32,
             np.dtype('int64'): DT_INT64}


# ---- crc32c (Castagnoli), masked as LevelDB / TensorFlow do -------------------------------------------
def _make_crc_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return tbl


_CRC_TABLE = _make_crc_table()


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in bytes(data):
        crc = tbl[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / tiny proto codec -----------------------------------------------------------------------
def _get_varint(buf, pos):
    result, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


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


def _parse_proto(buf):
    """-> list of (field_number, wire_type, value); value is int (varint/fixed) or bytes (length-delimited)."""
    out, pos = [], 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            if len(v) != ln:
                raise ValueError("truncated proto field")
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported proto wire type %d" % wt)
        out.append((fn, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def parse_bundle_entry(buf):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto)."""
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'sliced': False}
    for fn, wt, v in _parse_proto(buf):
        if fn == 1:
            e['dtype'] = v
        elif fn == 2:  # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1; } }
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e['shape'].append(size)
        elif fn == 3:
            e['shard_id'] = v
        elif fn == 4:
            e['offset'] = v
        elif fn == 5:
            e['size'] = v
        elif fn == 6:
            e['crc32c'] = v
        elif fn == 7:
            e['sliced'] = True
    return e


def _encode_bundle_entry(dtype, shape, offset, size, crc):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(s) for s in shape))
    out = b'\x08' + _put_varint(dtype)
    out += b'\x12' + _put_varint(len(dims)) + dims
    if offset:
        out += b'\x20' + _put_varint(offset)
    out += b'\x28' + _put_varint(size)
    out += b'\x35' + struct.pack('<I', crc)
    return out


# ---- snappy (raw format), for index blocks written with compression -----------------------------------
def _snappy_uncompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], 'little')
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block: length mismatch")
    return bytes(out)


# ---- SSTable ---------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    trailer = data[offset + size:offset + size + 5]
    if len(raw) != size or len(trailer) != 5:
        raise ValueError("truncated table block")
    if verify:
        want = struct.unpack('<I', trailer[1:5])[0]
        if mask_crc(crc32c(raw + trailer[:1])) != want:
            raise ValueError("table block checksum mismatch")
    if trailer[0] == 0:
        return raw
    if trailer[0] == 1:
        return _snappy_uncompress(raw)
    raise ValueError("unknown block compression type %d" % trailer[0])


def _block_entries(block):
    if len(block) < 4:
        raise ValueError("table block too small")
    num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise ValueError("corrupt restart array")
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key):
            raise ValueError("corrupt key prefix")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        value = bytes(block[pos:pos + vlen])
        pos += vlen
        yield key, value


def read_table(path, verify=True):
    """-> list of (key bytes, value bytes) of an SSTable file, in key order."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48:
        raise ValueError("%s: too small to be a table" % path)
    footer = data[-48:]
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise ValueError("%s: bad table magic (not a TensorBundle index)" % path)
    pos = 0
    _, pos = _get_varint(footer, pos)       # metaindex handle
    _, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size, verify)):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, off, size, verify)))
    return out


def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path, items, block_entries=64):
    """Writes sorted (key, value) pairs as an uncompressed SSTable."""
    items = sorted(items)
    blob = bytearray()
    index = []

    def emit(block):
        off = len(blob)
        blob.extend(block)
        blob.extend(b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return off, len(block)

    for i in range(0, max(len(items), 1), block_entries):
        chunk = items[i:i + block_entries]
        off, size = emit(_build_block(chunk))
        sep = chunk[-1][0] if chunk else b''
        index.append((sep, _put_varint(off) + _put_varint(size)))
    meta_off, meta_size = emit(_build_block([]))
    idx_off, idx_size = emit(_build_block(index, restart_interval=1))
    footer = _put_varint(meta_off) + _put_varint(meta_size) + _put_varint(idx_off) + _put_varint(idx_size)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    blob.extend(footer)
    with open(path, 'wb') as f:
        f.write(bytes(blob))


# ---- TensorBundle -----------------------------------------------------------------------------------------
class BundleReader(object):
    def __init__(self, prefix, verify=True):
        self.prefix = prefix
        index_path = prefix + '.index'
        if not os.path.exists(index_path):
            raise FileNotFoundError("no checkpoint index at %s" % index_path)
        self.entries = {}
        self.num_shards = 1
        for k, v in read_table(index_path, verify):
            if k == b'':
                for fn, _, val in _parse_proto(v):   # BundleHeaderProto
                    if fn == 1:
                        self.num_shards = val
                    elif fn == 2 and val != 0:
                        raise ValueError("big-endian bundles are not supported")
                continue
            self.entries[k.decode('utf-8')] = parse_bundle_entry(v)

    def keys(self):
        return sorted(self.entries)

    def shape(self, key):
        return tuple(self.entries[key]['shape'])

    def get_tensor(self, key, verify=False):
        e = self.entries[key]
        if e['sliced']:
            raise ValueError("%s: partitioned (sliced) variables are not supported" % key)
        if e['dtype'] not in _NP_OF_DT:
            raise ValueError("%s: dtype %d is not a plain numeric tensor" % (key, e['dtype']))
        path = '%s.data-%05d-of-%05d' % (self.prefix, e['shard_id'], self.num_shards)
        with open(path, 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise ValueError("%s: data file truncated" % key)
        if verify and mask_crc(crc32c(raw)) != e['crc32c']:
            raise ValueError("%s: data checksum mismatch" % key)
        arr = np.frombuffer(raw, dtype=_NP_OF_DT[e['dtype']])
        n = int(np.prod(e['shape'])) if e['shape'] else 1
        if arr.size != n:
            raise ValueError("%s: %d bytes do not match shape %r" % (key, e['size'], e['shape']))
        return arr.reshape(e['shape']).copy()


def write_bundle(prefix, tensors):
    """Writes {name: ndarray} as a one-shard TensorBundle (<prefix>.index + <prefix>.data-00000-of-00001)."""
    d = os.path.dirname(prefix)
    if d and not os.path.exists(d):
        os.makedirs(d)
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'   # num_shards = 1, version { producer: 1 }
    items = [(b'', header)]
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(tensors):
            arr = np.ascontiguousarray(tensors[name])
            dt = _DT_OF_NP.get(arr.dtype)
            if dt is None:
                raise ValueError("%s: unsupported dtype %s" % (name, arr.dtype))
            raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
            f.write(raw)
            items.append((name.encode('utf-8'),
                          _encode_bundle_entry(dt, arr.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', items)


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: reads the `checkpoint` state file (text proto) in `directory`."""
    state = os.path.join(directory, 'checkpoint')
    if not os.path.isfile(state):
        return None
    with open(state) as f:
        for line in f:
            m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"\s*$', line)
            if m:
                p = m.group(1)
                return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


def write_checkpoint_state(directory, name):
    with open(os.path.join(directory, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))


# ---- name maps ----------------------------------------------------------------------------------------------
def v1_names(scope, n_layers):
    """[(weights_name, biases_name)] for the n_layers dense layers of one model.Generator scope."""
    out = []
    for i in range(n_layers):
        fc = 'fully_connected' if i == 0 else 'fully_connected_%d' % i
        out.append(('%s/%s/weights' % (scope, fc), '%s/%s/biases' % (scope, fc)))
    return out


def v2_names(attr, n_layers):
    suffix = '/.ATTRIBUTES/VARIABLE_VALUE'
    return [('%s/all_layers/%d/0/kernel%s' % (attr, i, suffix), '%s/all_layers/%d/0/bias%s' % (attr, i, suffix))
            for i in range(n_layers)]


def _load_net(reader, net, names):
    ws, bs = [], []
    for wn, bn in names:
        if wn not in reader.entries or bn not in reader.entries:
            raise KeyError("checkpoint %s has no variable %s (have e.g. %s)"
                           % (reader.prefix, wn if wn not in reader.entries else bn, reader.keys()[:4]))
        ws.append(reader.get_tensor(wn))
        bs.append(reader.get_tensor(bn))
    net.set_weights(ws, bs)


def _v2_names_tolerant(reader, attr, net):
    """The expected v2 keys, or — the object-graph path being unverified — any '<attr>/...kernel' variables
    whose shapes chain through the net's widths, ordered by the numbers in their paths."""
    n = len(net.dims) - 1
    names = v2_names(attr, n)
    if all(w in reader.entries and b in reader.entries for w, b in names):
        return names
    suffix = '/.ATTRIBUTES/VARIABLE_VALUE'
    kernels = [k for k in reader.keys()
               if k.startswith(attr + '/') and k.endswith('kernel' + suffix) and '.OPTIMIZER_SLOT' not in k]
    kernels.sort(key=lambda k: [int(t) for t in re.findall(r'\d+', k)])
    want = list(zip(net.dims[:-1], net.dims[1:]))
    if [reader.shape(k) for k in kernels] != want:
        raise KeyError("checkpoint %s: cannot find the %d dense layers of %s (kernels found: %r)"
                       % (reader.prefix, n, attr, kernels))
    return [(k, k[:-len('kernel' + suffix)] + 'bias' + suffix) for k in kernels]


def load_v1(prefix, g_net, h_net, verify=False):
    """tf.train.Saver checkpoint of the v1 model (main_density_est.py:402,410). Adam slots are ignored."""
    r = BundleReader(prefix)
    _load_net(r, g_net, v1_names(g_net.name, len(g_net.dims) - 1))
    _load_net(r, h_net, v1_names(h_net.name, len(h_net.dims) - 1))


def load_v2(prefix, g_net, e_net):
    """tf.train.Checkpoint of the v2 model (roundtrip.py:48-57)."""
    r = BundleReader(prefix)
    _load_net(r, g_net, _v2_names_tolerant(r, 'g_net', g_net))
    _load_net(r, e_net, _v2_names_tolerant(r, 'e_net', e_net))


def _net_tensors(net, names):
    out = {}
    for (wn, bn), w, b in zip(names, net.weights, net.biases):
        out[wn] = w
        out[bn] = b
    return out


def save_v1(prefix, g_net, h_net):
    t = _net_tensors(g_net, v1_names(g_net.name, len(g_net.dims) - 1))
    t.update(_net_tensors(h_net, v1_names(h_net.name, len(h_net.dims) - 1)))
    write_bundle(prefix, t)


def save_v2(prefix, g_net, e_net):
    t = _net_tensors(g_net, v2_names('g_net', len(g_net.dims) - 1))
    t.update(_net_tensors(e_net, v2_names('e_net', len(e_net.dims) - 1)))
    write_bundle(prefix, t)
