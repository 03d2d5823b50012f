"""Reader (and a minimal writer) for the TensorFlow checkpoint files the reference saves.

The reference stores G / H weights with `tf.train.Saver` (v1: /root/reference/main_density_est.py:123,
389-411) or `tf.train.Checkpoint` + `CheckpointManager` (v2: /root/reference/src/pyroundtrip/roundtrip.py:
48-58).  Both write the TensorBundle format: `<prefix>.index`, an SSTable (LevelDB table format) mapping
tensor name -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}, and
`<prefix>.data-00000-of-00001` holding the raw little-endian tensor bytes.

No TensorFlow, no protobuf package: the SSTable blocks, the small protos (BundleHeaderProto, BundleEntryProto,
TrackableObjectGraph) and (optional) snappy blocks are decoded here by hand.  PARITY UNPINNED: not verified against a
TF-written file — none exists offline; the writer below produces the same layout so that tests can
round-trip and fixtures can be exchanged with a TensorFlow install elsewhere, and tests/test_abi_host.py also feeds the
reader an index assembled byte by byte by independent code (snappy blocks, restart interval 16, Adam slots, a string
tensor, two data shards).

Variable names:
  v1  <scope>/fully_connected/weights, <scope>/fully_connected/biases, <scope>/fully_connected_<i>/...
      for i = 1..nb_layers (tcl.fully_connected's default scopes under tf.variable_scope(name), model.py:63)
  v2  tf.train.Checkpoint stores the object graph it saved in the string tensor `_CHECKPOINTABLE_OBJECT_GRAPH`
      (TrackableObjectGraph: nodes -> children (local_name, node_id) / attributes (name, checkpoint_key)).  The
      reader walks it along the attribute path the CODE defines — root -> g_net -> all_layers -> <i> -> 0 -> kernel / bias
      (`Checkpoint(g_net=...)`, roundtrip.py:48; `self.all_layers.append([fc_layer, norm_layer])`,
      src/pyroundtrip/model.py:25) — and reads whatever checkpoint_key the VARIABLE_VALUE attribute names.  Only when a
      file has no object graph does it fall back to the expected key text
      <attr>/all_layers/<i>/0/kernel/.ATTRIBUTES/VARIABLE_VALUE (or any kernels under <attr>/ whose shapes chain).
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


def _crc32c_py(data, crc=0):
    crc ^= 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in bytes(data):
        crc = tbl[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


_crc_native = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of `data`, continuing from `crc`.  Long inputs (the per-tensor checksums of a 12 MB
    checkpoint) go through rt_crc32c of librt_is.so when the library is there (host code, no GPU); the pure-Python loop
    gives the same values (tests compare them) and serves everything else."""
    global _crc_native
    data = bytes(data)
    if len(data) >= 4096 and _crc_native is not False:
        if _crc_native is None:
            try:
                import ctypes
                from . import _lib
                fn = _lib.load().rt_crc32c
                fn.restype = ctypes.c_uint32
                fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
                _crc_native = fn
            except (ImportError, OSError, AttributeError):
                _crc_native = False
        if _crc_native:
            return int(_crc_native(data, len(data), crc))
    return _crc32c_py(data, crc)


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


def _encode_bundle_entry(dtype, shape, offset, size, crc, shard_id=0):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(s) for s in shape))
    out = b'\x08' + _put_varint(dtype)
    out += b'\x12' + _put_varint(len(dims)) + dims
    if shard_id:
        out += b'\x18' + _put_varint(shard_id)
    if offset:
        out += b'\x20' + _put_varint(offset)
    out += b'\x28' + _put_varint(size)
    out += b'\x35' + struct.pack('<I', crc)
    return out


# ---- TrackableObjectGraph (tensorflow/core/protobuf/trackable_object_graph.proto) ------------------------------
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'


def parse_object_graph(buf):
    """-> list of nodes, node i = {'children': {local_name: node_id}, 'attributes': {name: checkpoint_key},
    'slots': [(original_variable_node_id, slot_name, slot_variable_node_id)]}.
    TrackableObjectGraph { repeated TrackableObject nodes = 1 }; TrackableObject { repeated ObjectReference children = 1
    {node_id = 1, local_name = 2}; repeated SerializedTensor attributes = 2 {name = 1, full_name = 2, checkpoint_key = 3};
    repeated SlotVariableReference slot_variables = 3 {original_variable_node_id = 1, slot_name = 2,
    slot_variable_node_id = 3} }."""
    nodes = []
    for fn, wt, v in _parse_proto(buf):
        if fn != 1 or wt != 2:
            continue
        node = {'children': {}, 'attributes': {}, 'slots': []}
        for f2, w2, v2 in _parse_proto(v):
            if w2 != 2:
                continue
            sub = _parse_proto(v2)
            if f2 == 1:
                nid, name = 0, ''
                for f3, _, v3 in sub:
                    if f3 == 1:
                        nid = v3
                    elif f3 == 2:
                        name = v3.decode('utf-8')
                node['children'][name] = nid
            elif f2 == 2:
                name, key = '', ''
                for f3, _, v3 in sub:
                    if f3 == 1:
                        name = v3.decode('utf-8')
                    elif f3 == 3:
                        key = v3.decode('utf-8')
                node['attributes'][name] = key
            elif f2 == 3:
                orig, slot, var = 0, '', 0
                for f3, _, v3 in sub:
                    if f3 == 1:
                        orig = v3
                    elif f3 == 2:
                        slot = v3.decode('utf-8')
                    elif f3 == 3:
                        var = v3
                node['slots'].append((orig, slot, var))
        nodes.append(node)
    return nodes


def _encode_object_graph(nodes):
    def field(fn, payload):
        return _put_varint((fn << 3) | 2) + _put_varint(len(payload)) + payload
    out = b''
    for node in nodes:
        body = b''
        for name, nid in node.get('children', {}).items():
            body += field(1, (b'\x08' + _put_varint(nid) if nid else b'') + field(2, name.encode('utf-8')))
        for name, key in node.get('attributes', {}).items():
            body += field(2, field(1, name.encode('utf-8')) + field(2, key.split('/.ATTRIBUTES')[0].encode('utf-8'))
                          + field(3, key.encode('utf-8')))
        for orig, slot, var in node.get('slots', []):
            body += field(3, b'\x08' + _put_varint(orig) + field(2, slot.encode('utf-8')) + b'\x18' + _put_varint(var))
        out += field(1, body)
    return out


def resolve_path(nodes, path, attribute='VARIABLE_VALUE'):
    """checkpoint_key of `attribute` on the node reached from the root (node 0) along the child names in `path`."""
    nid = 0
    for i, name in enumerate(path):
        kids = nodes[nid]['children']
        if name not in kids:
            raise KeyError("object graph: no child %r under /%s (children: %s)"
                           % (name, '/'.join(path[:i]), sorted(kids)))
        nid = kids[name]
        if not 0 <= nid < len(nodes):
            raise ValueError("object graph: child %r points at node %d of %d" % (name, nid, len(nodes)))
    attrs = nodes[nid]['attributes']
    if attribute not in attrs:
        raise KeyError("object graph: /%s has no %s attribute (has %s)" % ('/'.join(path), attribute, sorted(attrs)))
    return attrs[attribute]


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

    def _raw(self, key, verify):
        e = self.entries[key]
        path = '%s.data-%05d-of-%05d' % (self.prefix, e['shard_id'], self.num_shards)
        with open(path, 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise ValueError("%s: data file %s truncated" % (key, os.path.basename(path)))
        return raw

    def get_strings(self, key, verify=True):
        """The elements (bytes) of a DT_STRING tensor.  On disk (tensor_bundle.cc, WriteStringTensor): the elements'
        lengths as varint64s, a 4-byte masked crc32c of those lengths (taken over each length as a little-endian
        uint32, or uint64 beyond 2^32), then the elements' bytes; the entry's crc32c covers the lengths in that
        fixed-width form, the 4 checksum bytes and the elements' bytes."""
        e = self.entries[key]
        if e['dtype'] != DT_STRING:
            raise ValueError("%s: not a string tensor" % key)
        raw = self._raw(key, verify)
        n = int(np.prod(e['shape'])) if e['shape'] else 1
        pos, lens = 0, []
        for _ in range(n):
            ln, pos = _get_varint(raw, pos)
            lens.append(ln)
        fixed = b''.join(struct.pack('<I', ln) if ln <= 0xFFFFFFFF else struct.pack('<Q', ln) for ln in lens)
        cks = raw[pos:pos + 4]
        pos += 4
        if verify:
            if len(cks) != 4 or struct.unpack('<I', cks)[0] != mask_crc(crc32c(fixed)):
                raise ValueError("%s: string-length checksum mismatch" % key)
            if mask_crc(crc32c(raw[pos:], crc32c(cks, crc32c(fixed)))) != e['crc32c']:
                raise ValueError("%s: data checksum mismatch" % key)
        out = []
        for ln in lens:
            out.append(raw[pos:pos + ln])
            pos += ln
        if pos != len(raw):
            raise ValueError("%s: string tensor size mismatch" % key)
        return out

    def object_graph(self):
        """The TrackableObjectGraph a tf.train.Checkpoint stored (list of nodes, see parse_object_graph), or None."""
        if OBJECT_GRAPH_KEY not in self.entries:
            return None
        # (the string tensor's own checksums are not insisted on here: its on-disk layout is restated from
        # tensor_bundle.cc without a TF-written file to confirm it, and a damaged graph fails loudly in resolve_path)
        return parse_object_graph(self.get_strings(OBJECT_GRAPH_KEY, verify=False)[0])

    def get_tensor(self, key, verify=True):
        e = self.entries[key]
        if e['sliced']:
            raise ValueError("%s: partitioned (sliced) variables are not supported" % key)
        if e['dtype'] not in _NP_OF_DT:
            raise ValueError("%s: dtype %d is not a plain numeric tensor" % (key, e['dtype']))
        raw = self._raw(key, verify)
        if verify and mask_crc(crc32c(raw)) != e['crc32c']:
            raise ValueError("%s: data checksum mismatch" % key)
        arr = np.frombuffer(raw, dtype=_NP_OF_DT[e['dtype']])
        n = int(np.prod(e['shape'])) if e['shape'] else 1
        if arr.size != n:
            raise ValueError("%s: %d bytes do not match shape %r" % (key, e['size'], e['shape']))
        return arr.reshape(e['shape']).copy()


def write_bundle(prefix, tensors, shards=1):
    """Writes {name: ndarray | bytes} as a TensorBundle: <prefix>.index + <prefix>.data-0000k-of-0000n, tensors dealt
    round-robin over `shards` data files (a bytes value becomes a scalar DT_STRING tensor)."""
    d = os.path.dirname(prefix)
    if d and not os.path.exists(d):
        os.makedirs(d)
    header = b'\x08' + _put_varint(shards) + b'\x1a\x02\x08\x01'   # num_shards, version { producer: 1 }
    items = [(b'', header)]
    files = [open('%s.data-%05d-of-%05d' % (prefix, k, shards), 'wb') for k in range(shards)]
    offsets = [0] * shards
    try:
        for i, name in enumerate(sorted(tensors)):
            k = i % shards
            val = tensors[name]
            if isinstance(val, (bytes, bytearray)):
                val = bytes(val)
                fixed = struct.pack('<I', len(val))
                cks = struct.pack('<I', mask_crc(crc32c(fixed)))
                raw = _put_varint(len(val)) + cks + val
                crc = crc32c(val, crc32c(cks, crc32c(fixed)))
                dt, shape = DT_STRING, ()
            else:
                arr = np.ascontiguousarray(val)
                dt = _DT_OF_NP.get(arr.dtype)
                if dt is None:
                    raise ValueError("%s: unsupported dtype %s" % (name, arr.dtype))
                raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
                crc, shape = crc32c(raw), arr.shape
            files[k].write(raw)
            items.append((name.encode('utf-8'),
                          _encode_bundle_entry(dt, shape, offsets[k], len(raw), mask_crc(crc), shard_id=k)))
            offsets[k] += len(raw)
    finally:
        for f in files:
            f.close()
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


def _load_net(reader, net, names, verify=True):
    ws, bs = [], []
    for wn, bn in names:
        if wn not in reader.entries or bn not in reader.entries:
            raise KeyError("checkpoint %s has no variable %s (have e.g. %s)"
                           % (reader.prefix, wn if wn not in reader.entries else bn, reader.keys()[:4]))
        ws.append(reader.get_tensor(wn, verify))
        bs.append(reader.get_tensor(bn, verify))
    net.set_weights(ws, bs)


def _v2_names_from_graph(nodes, attr, n_layers):
    """[(kernel key, bias key)] of net `attr`'s dense layers, read off the object graph along the path the code defines:
    Checkpoint(<attr>=net) (roundtrip.py:48) -> net.all_layers (a list) -> [i] (the [fc_layer, norm_layer] pair,
    src/pyroundtrip/model.py:25) -> [0] (the Dense layer) -> kernel / bias."""
    return [(resolve_path(nodes, [attr, 'all_layers', str(i), '0', 'kernel']),
             resolve_path(nodes, [attr, 'all_layers', str(i), '0', 'bias'])) for i in range(n_layers)]


def _v2_names_tolerant(reader, attr, net):
    """For a file WITHOUT an object graph: the expected v2 key text, or any '<attr>/...kernel' variables whose shapes
    chain through the net's widths, ordered by the numbers in their paths."""
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


def load_v1(prefix, g_net, h_net, verify=True):
    """tf.train.Saver checkpoint of the v1 model (main_density_est.py:402,410). Adam slots are ignored.
    verify: check the index blocks' and every loaded tensor's crc32c (a corrupted data file raises)."""
    r = BundleReader(prefix, verify)
    _load_net(r, g_net, v1_names(g_net.name, len(g_net.dims) - 1), verify)
    _load_net(r, h_net, v1_names(h_net.name, len(h_net.dims) - 1), verify)


def load_v2(prefix, g_net, e_net, verify=True):
    """tf.train.Checkpoint of the v2 model (roundtrip.py:48-57): keys resolved through the stored object graph."""
    r = BundleReader(prefix, verify)
    nodes = r.object_graph()
    for attr, net in (('g_net', g_net), ('e_net', e_net)):
        if nodes is not None:
            names = _v2_names_from_graph(nodes, attr, len(net.dims) - 1)
        else:
            names = _v2_names_tolerant(r, attr, net)
        _load_net(r, net, names, verify)


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


def v2_object_graph(nets):
    """The TrackableObjectGraph nodes for Checkpoint(**nets) with the key text of v2_names: root -> <attr> ->
    all_layers -> <i> -> 0 -> kernel / bias, each variable node carrying a VARIABLE_VALUE attribute."""
    nodes = [{'children': {}}]

    def add(parent, name):
        nodes.append({'children': {}, 'attributes': {}})
        nodes[parent]['children'][name] = len(nodes) - 1
        return len(nodes) - 1

    for attr, net in nets:
        a = add(0, attr)
        lst = add(a, 'all_layers')
        for i, (wn, bn) in enumerate(v2_names(attr, len(net.dims) - 1)):
            pair = add(lst, str(i))
            dense = add(pair, '0')
            nodes[add(dense, 'kernel')]['attributes']['VARIABLE_VALUE'] = wn
            nodes[add(dense, 'bias')]['attributes']['VARIABLE_VALUE'] = bn
    return nodes


def save_v2(prefix, g_net, e_net, shards=1):
    t = _net_tensors(g_net, v2_names('g_net', len(g_net.dims) - 1))
    t.update(_net_tensors(e_net, v2_names('e_net', len(e_net.dims) - 1)))
    t[OBJECT_GRAPH_KEY] = _encode_object_graph(v2_object_graph([('g_net', g_net), ('e_net', e_net)]))
    write_bundle(prefix, t, shards=shards)
