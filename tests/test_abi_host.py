"""CPU tests (no GPU, no compute calls): the C-ABI library loads and exports every symbol include/rt_is.h
declares; the host-side mirror of the reference interface (nets, model objects, checkpoint reader,
sharding) behaves like the reference's."""
import os
import re
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from roundtrip_amd import _lib, checkpoint as ck, dist as rdist
from roundtrip_amd import BaseFullyConnectedNet, Generator, Roundtrip, RoundtripModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ----------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt_is.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_no_silent_cpu_fallback_without_gpu():
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(_lib.RtError):
        _lib.Engine(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under roundtrip_amd/ may import, link or execute it."""
    pkg = os.path.join(ROOT, "roundtrip_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, fn)


# ---- nets / model objects -------------------------------------------------------------------------------
def test_generator_shapes_follow_reference_constructors():
    g = Generator(2, 3, "g_net", nb_layers=10, nb_units=512, rng=np.random.RandomState(0))   # model.py:54
    assert g.dims == [2] + [512] * 10 + [3]
    assert [w.shape for w in g.weights][:2] == [(2, 512), (512, 512)] and g.weights[-1].shape == (512, 3)
    assert all(np.all(b == 0) for b in g.biases)                      # zero-initialised biases
    lim = np.sqrt(6.0 / (512 + 512))
    assert np.abs(g.weights[3]).max() <= lim                           # Glorot uniform
    e = BaseFullyConnectedNet(3, 2, "e_net", nb_units=[256] * 10, rng=np.random.RandomState(0))  # v2 model.py:7
    assert e.dims == [3] + [256] * 10 + [2]
    with pytest.raises(NotImplementedError):
        BaseFullyConnectedNet(3, 2, "e_net", nb_units=[8], batchnorm=True)
    with pytest.raises(ValueError):
        BaseFullyConnectedNet(3, 2, "e_net", nb_units=[])
    with pytest.raises(ValueError):
        g.set_weights(g.weights[:-1], g.biases[:-1])
    bad = [w.copy() for w in g.weights]
    bad[2] = bad[2][:, :100]
    with pytest.raises(ValueError):
        g.set_weights(bad, g.biases)
    v = g.version
    g.set_weights(g.weights, g.biases)
    assert g.version == v + 1


def _params(tmp_path, **kw):
    p = dict(dataset="indep_gmm", output_dir=str(tmp_path), x_dim=2, z_dim=2, lr=2e-4, alpha=10, beta=1, gamma=0,
             g_d_freq=1, g_units=[128] * 3, e_units=[128] * 3, dz_units=[128, 128], dx_units=[256] * 4,
             save_model=False, sd_x=0.05, scale=0.5, sample_size=20000)
    p.update(kw)
    return p


def test_roundtrip_v2_constructor_mirrors_reference(tmp_path):
    m = Roundtrip(_params(tmp_path), timestamp="t0", random_seed=123)          # roundtrip.py:12
    assert m.g_net.input_dim == 2 and m.g_net.output_dim == 2 and m.e_net.nb_units == [128] * 3
    assert m.checkpoint_path == "%s/checkpoints/indep_gmm/t0" % tmp_path        # roundtrip.py:36
    assert m.save_dir == "%s/results/indep_gmm/t0" % tmp_path and os.path.isdir(m.save_dir)   # :42-46
    assert not os.path.exists(m.checkpoint_path)                               # save_model False
    assert m.get_config() == {"params": m.params}
    m2 = Roundtrip(_params(tmp_path), timestamp="t0", random_seed=123)
    assert all(np.array_equal(a, b) for a, b in zip(m.g_net.get_weights(), m2.g_net.get_weights()))
    m3 = Roundtrip(_params(tmp_path, save_model=True), timestamp="t1")
    assert os.path.isdir(m3.checkpoint_path)
    # signatures kept verbatim
    import inspect
    sig = inspect.signature(Roundtrip.estimate_px_with_IS)
    assert list(sig.parameters)[1:] == ["x_points", "sd_x", "scale", "sample_size", "bs", "df", "log"]
    assert [sig.parameters[k].default for k in list(sig.parameters)[2:]] == [0.5, 0.5, 40000, 1024, 1, True]
    sig = inspect.signature(RoundtripModel.estimate_py_with_IS)
    assert list(sig.parameters)[1:] == ["y_points", "epoch", "sd_y", "scale", "sample_size", "bs", "log", "save"]
    assert [sig.parameters[k].default for k in list(sig.parameters)[3:]] == [0.5, 0.5, 40000, 1024, True, True]
    with pytest.raises(AssertionError):            # reference: assert x.shape[-1] == self.x_dim
        m.estimate_px_with_IS(np.zeros((4, 3)))


def test_roundtrip_v2_restores_latest_checkpoint(tmp_path, capsys):
    src = Roundtrip(_params(tmp_path), timestamp="src", random_seed=1)
    ckdir = "%s/checkpoints/indep_gmm/run" % tmp_path
    os.makedirs(ckdir)
    ck.save_v2(os.path.join(ckdir, "ckpt-3"), src.g_net, src.e_net)
    ck.write_checkpoint_state(ckdir, "ckpt-3")
    m = Roundtrip(_params(tmp_path), timestamp="run", random_seed=2)          # roundtrip.py:56-58
    assert "Latest checkpoint restored!!" in capsys.readouterr().out
    assert all(np.array_equal(a, b) for a, b in zip(src.g_net.get_weights(), m.g_net.get_weights()))
    assert all(np.array_equal(a, b) for a, b in zip(src.e_net.get_weights(), m.e_net.get_weights()))


def test_roundtrip_model_v1(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    g = Generator(3, 6, "g_net", 3, 128, rng=np.random.RandomState(0))
    h = Generator(6, 3, "h_net", 3, 128, rng=np.random.RandomState(1))
    m = RoundtripModel(g, h, None, None, None, None, "arem", None, 64, 10.0, 10.0, 1, False)   # :30
    assert (m.x_dim, m.y_dim, m.df) == (3, 6, 1)
    assert m.save_dir.startswith("data/density_est_") and m.save_dir.endswith("_arem_x_dim=3_y_dim=6_alpha=10.0_beta=10.0")
    with pytest.raises(ValueError):
        RoundtripModel(g, Generator(5, 3, "h_net", 3, 128))
    with pytest.raises(AssertionError):
        m.estimate_py_with_IS(np.zeros((4, 5)), 0)
    # v1 load(): checkpoint/density_est_<ts>_<data>_x_dim=..  (main_density_est.py:409-410)
    d = "checkpoint/density_est_TS_arem_x_dim=3_y_dim=6_alpha=10.0_beta=10.0"
    g2 = Generator(3, 6, "g_net", 3, 128, rng=np.random.RandomState(5))
    h2 = Generator(6, 3, "h_net", 3, 128, rng=np.random.RandomState(6))
    ck.save_v1(os.path.join(d, "model.ckpt-7"), g2, h2)
    m.load(timestamp="TS", epoch=7)
    assert all(np.array_equal(a, b) for a, b in zip(g.get_weights(), g2.get_weights()))
    assert all(np.array_equal(a, b) for a, b in zip(h.get_weights(), h2.get_weights()))


# ---- checkpoint format -----------------------------------------------------------------------------------
def test_crc32c_known_answers():
    assert ck.crc32c(b"123456789") == 0xE3069283            # the standard CRC-32C check value
    assert ck.crc32c(b"\x00" * 32) == 0x8A9136AA            # RFC 3720 B.4
    assert ck.mask_crc(0) == 0xa282ead8


def test_varint_and_proto_roundtrip():
    for v in (0, 1, 127, 128, 300, 2 ** 32, 2 ** 63 - 1):
        enc = ck._put_varint(v)
        assert ck._get_varint(enc, 0) == (v, len(enc))
    e = ck.parse_bundle_entry(ck._encode_bundle_entry(ck.DT_FLOAT, (512, 2), 4096, 4096, 0xdeadbeef))
    assert e["dtype"] == 1 and e["shape"] == [512, 2] and e["offset"] == 4096 and e["size"] == 4096
    assert e["crc32c"] == 0xdeadbeef and not e["sliced"]


def test_sstable_multi_block_prefix_compression(tmp_path):
    items = [(("scope/var_%04d/weights" % i).encode(), os.urandom(1 + i % 7)) for i in range(500)]
    items.append((b"", b"hdr"))
    path = str(tmp_path / "t.index")
    ck.write_table(path, items, block_entries=37)
    assert ck.read_table(path) == sorted(items)
    raw = bytearray(open(path, "rb").read())
    raw[10] ^= 0xFF
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        ck.read_table(path)
    with pytest.raises(ValueError):
        open(path, "wb").write(b"\x00" * 100)
        ck.read_table(path)


def test_snappy_block_decoding():
    # literal "abcd" + copy(offset 4, len 8) + literal "xyz"  -> "abcdabcdabcdxyz"
    raw = bytes([15]) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([2 << 2]) + b"xyz"
    assert ck._snappy_uncompress(raw) == b"abcdabcdabcdxyz"
    with pytest.raises(ValueError):
        ck._snappy_uncompress(bytes([4]) + bytes([((8 - 4) << 2) | 1, 9]))


def test_bundle_roundtrip_and_names(tmp_path):
    g = Generator(2, 2, "g_net", 10, 64, rng=np.random.RandomState(0))
    h = Generator(2, 2, "h_net", 10, 32, rng=np.random.RandomState(1))
    prefix = str(tmp_path / "model.ckpt-12")
    ck.save_v1(prefix, g, h)
    r = ck.BundleReader(prefix)
    # variable names of tcl.fully_connected under tf.variable_scope(name)
    assert "g_net/fully_connected/weights" in r.entries and "h_net/fully_connected_10/biases" in r.entries
    assert len(r.keys()) == 2 * 2 * 11
    assert r.shape("g_net/fully_connected_10/weights") == (64, 2)
    w = r.get_tensor("g_net/fully_connected_3/weights", verify=True)
    assert np.array_equal(w, g.weights[3])
    # Adam slots and other variables in the file are ignored by the loader
    extra = {"g_net/fully_connected/weights/Adam": np.zeros((2, 64), np.float32), "beta1_power": np.float32(0.5)}
    t = ck._net_tensors(g, ck.v1_names("g_net", 11))
    t.update(ck._net_tensors(h, ck.v1_names("h_net", 11)))
    t.update(extra)
    ck.write_bundle(prefix, t)
    g2 = Generator(2, 2, "g_net", 10, 64)
    h2 = Generator(2, 2, "h_net", 10, 32)
    ck.load_v1(prefix, g2, h2)
    assert all(np.array_equal(a, b) for a, b in zip(g.get_weights(), g2.get_weights()))
    with pytest.raises(KeyError):
        ck.load_v1(prefix, Generator(2, 2, "g_net", 11, 64), h2)       # one layer too many
    with pytest.raises(FileNotFoundError):
        ck.BundleReader(str(tmp_path / "nope"))
    # data checksum is checked on request
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[5] ^= 1
    open(data, "wb").write(bytes(raw))
    r = ck.BundleReader(prefix)
    bad = 0
    for k in r.keys():
        try:
            r.get_tensor(k, verify=True)
        except ValueError:
            bad += 1
    assert bad == 1


def test_v2_object_graph_keys_and_tolerant_lookup(tmp_path):
    g = BaseFullyConnectedNet(2, 2, "g_net", [16, 16], rng=np.random.RandomState(0))
    e = BaseFullyConnectedNet(2, 2, "e_net", [8, 8], rng=np.random.RandomState(1))
    prefix = str(tmp_path / "ckpt-1")
    ck.save_v2(prefix, g, e)
    assert "g_net/all_layers/0/0/kernel/.ATTRIBUTES/VARIABLE_VALUE" in ck.BundleReader(prefix).entries
    # a different object-graph spelling with the same shapes is still found (the v2 path is unverified)
    sfx = "/.ATTRIBUTES/VARIABLE_VALUE"
    t = {}
    for attr, net in (("g_net", g), ("e_net", e)):
        for i, (w, b) in enumerate(zip(net.weights, net.biases)):
            t["%s/layer_with_weights-%d/kernel%s" % (attr, i, sfx)] = w
            t["%s/layer_with_weights-%d/bias%s" % (attr, i, sfx)] = b
            t["%s/layer_with_weights-%d/kernel/.OPTIMIZER_SLOT/g_e_optimizer/m%s" % (attr, i, sfx)] = w * 0
    ck.write_bundle(prefix, t)
    g2 = BaseFullyConnectedNet(2, 2, "g_net", [16, 16])
    e2 = BaseFullyConnectedNet(2, 2, "e_net", [8, 8])
    ck.load_v2(prefix, g2, e2)
    assert all(np.array_equal(a, b) for a, b in zip(g.get_weights(), g2.get_weights()))
    assert all(np.array_equal(a, b) for a, b in zip(e.get_weights(), e2.get_weights()))
    with pytest.raises(KeyError):
        ck.load_v2(prefix, BaseFullyConnectedNet(2, 2, "g_net", [16, 16, 16]), e2)


# ---- a TensorBundle assembled byte by byte by code that shares nothing with roundtrip_amd/checkpoint.py --------------
def _vi(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def _crc32c_bitwise(data, crc=0):
    crc ^= 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def _masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _ld(field, payload):                       # length-delimited proto field
    return _vi((field << 3) | 2) + _vi(len(payload)) + payload


def _snappy_greedy(raw):
    """A valid raw-snappy stream for `raw`: copies (2-byte offsets, <= 64 bytes) where an earlier occurrence of the next
    >= 4 bytes exists, literals (short and 60/61-tagged long ones) elsewhere."""
    out, lit, i = _vi(len(raw)), b"", 0

    def flush(lit):
        if not lit:
            return b""
        n = len(lit) - 1
        if n < 60:
            return bytes([n << 2]) + lit
        if n < 256:
            return bytes([60 << 2, n]) + lit
        return bytes([61 << 2, n & 0xFF, n >> 8]) + lit

    while i < len(raw):
        best, off = 0, 0
        if i + 4 <= len(raw):
            j = raw.rfind(raw[i:i + 4], max(0, i - 60000), i + 3)
            if 0 <= j < i:
                ln = 4
                while ln < 64 and i + ln < len(raw) and raw[j + ln] == raw[i + ln]:   # may overlap: snappy allows it
                    ln += 1
                best, off = ln, i - j
        if best:
            out += flush(lit)
            lit = b""
            out += bytes([((best - 1) << 2) | 2, off & 0xFF, off >> 8])
            i += best
        else:
            lit += raw[i:i + 1]
            i += 1
            if len(lit) == 300:
                out += flush(lit)
                lit = b""
    return out + flush(lit)


def _hand_table(items, per_block, compress):
    """LevelDB table: data blocks of `per_block` entries with prefix compression and a restart point every 16 entries,
    an empty metaindex block, an index block, the 48-byte footer."""
    items = sorted(items)
    blob, index = b"", []

    def block(entries, interval):
        out, restarts, last = b"", [], b""
        for n, (k, v) in enumerate(entries):
            shared = 0
            if n % interval == 0:
                restarts.append(len(out))
            else:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
            last = k
        for r in restarts or [0]:
            out += r.to_bytes(4, "little")
        return out + len(restarts or [0]).to_bytes(4, "little")

    def emit(raw, compress):
        nonlocal blob
        body, kind = (_snappy_greedy(raw), 1) if compress else (raw, 0)
        handle = _vi(len(blob)) + _vi(len(body))
        blob += body + bytes([kind]) + _masked(_crc32c_bitwise(body + bytes([kind]))).to_bytes(4, "little")
        return handle

    for i in range(0, len(items), per_block):
        chunk = items[i:i + per_block]
        index.append((chunk[-1][0], emit(block(chunk, 16), compress)))
    meta = emit(block([], 16), False)
    idx = emit(block(index, 1), False)
    footer = meta + idx
    return blob + footer + b"\x00" * (40 - len(footer)) + (0xdb4775248b80fb57).to_bytes(8, "little")


def _hand_entry(dtype, shape, shard, offset, size, crc):
    dims = b"".join(_ld(2, b"\x08" + _vi(d)) for d in shape)
    e = b"\x08" + _vi(dtype) + _ld(2, dims)
    if shard:
        e += b"\x18" + _vi(shard)
    if offset:
        e += b"\x20" + _vi(offset)
    return e + b"\x28" + _vi(size) + b"\x35" + crc.to_bytes(4, "little")


def _hand_bundle(prefix, tensors, shards, per_block=40, compress=True):
    """tensors: {name: float32/int64 ndarray | bytes}; dealt over `shards` data files in name order, hash-like."""
    data = [b""] * shards
    items = [(b"", b"\x08" + _vi(shards) + _ld(3, b"\x08\x01"))]        # BundleHeaderProto: num_shards, version
    for n, name in enumerate(sorted(tensors)):
        k = (n * 7 + 3) % shards
        val = tensors[name]
        if isinstance(val, bytes):
            fixed = len(val).to_bytes(4, "little")
            cks = _masked(_crc32c_bitwise(fixed)).to_bytes(4, "little")
            raw, dtype, shape = _vi(len(val)) + cks + val, 7, ()
            crc = _crc32c_bitwise(fixed + cks + val)
        else:
            raw, dtype, shape = val.tobytes(), {"float32": 1, "int64": 9}[str(val.dtype)], val.shape
            crc = _crc32c_bitwise(raw)
        items.append((name.encode(), _hand_entry(dtype, shape, k, len(data[k]), len(raw), _masked(crc))))
        data[k] += raw
    for k in range(shards):
        open("%s.data-%05d-of-%05d" % (prefix, k, shards), "wb").write(data[k])
    open(prefix + ".index", "wb").write(_hand_table(items, per_block, compress))


def _keras_like_checkpoint(g, e):
    """What tf.train.Checkpoint(g_net=..., e_net=..., g_e_optimizer=...) plausibly writes for two Keras models whose
    dense layers hang off `all_layers` lists: variable keys spelled the way Keras' own layer tracking would spell them
    (NOT the all_layers/<i>/0 text), Adam slots, a save counter, and the object graph that ties the code's attribute
    path to those keys.  Returns (tensors, graph bytes)."""
    sfx = "/.ATTRIBUTES/VARIABLE_VALUE"
    tensors, nodes = {}, [dict(children=[], attrs=[], slots=[])]

    def add(parent, name):
        nodes.append(dict(children=[], attrs=[], slots=[]))
        nodes[parent]["children"].append((name, len(nodes) - 1))
        return len(nodes) - 1

    opt = add(0, "g_e_optimizer")
    ctr = add(0, "save_counter")
    nodes[ctr]["attrs"].append(("VARIABLE_VALUE", "save_counter", "save_counter" + sfx))
    tensors["save_counter" + sfx] = np.array(3, dtype=np.int64)
    for attr, net in (("e_net", e), ("g_net", g)):
        a = add(0, attr)
        lst = add(a, "all_layers")
        for i, (w, b) in enumerate(zip(net.weights, net.biases)):
            pair = add(lst, str(i))
            dense = add(pair, "0")
            nodes[a]["children"].append(("layer_with_weights-%d" % i, dense))        # a second path to the same node
            norm = add(pair, "1")                                                      # the BatchNormalization twin
            for var, arr in (("kernel", w), ("bias", b)):
                key = "%s/layer_with_weights-%d/%s%s" % (attr, i, var, sfx)
                v = add(dense, var)
                nodes[v]["attrs"].append(("VARIABLE_VALUE", "%s/dense_%d/%s" % (attr, i, var), key))
                tensors[key] = arr
                for slot in ("m", "v"):
                    sk = "%s/layer_with_weights-%d/%s/.OPTIMIZER_SLOT/g_e_optimizer/%s%s" % (attr, i, var, slot, sfx)
                    sv = len(nodes)
                    nodes.append(dict(children=[], attrs=[("VARIABLE_VALUE", "Adam/" + slot, sk)], slots=[]))
                    nodes[opt]["slots"].append((v, slot, sv))
                    tensors[sk] = np.zeros_like(arr) + (1 if slot == "m" else 2)
            nodes[norm]["attrs"].append(("VARIABLE_VALUE", "bn/gamma", "%s/bn%d/gamma%s" % (attr, i, sfx)))
            tensors["%s/bn%d/gamma%s" % (attr, i, sfx)] = np.ones(w.shape[1], np.float32)
    graph = b""
    for nd in nodes:
        body = b""
        for name, nid in nd["children"]:
            body += _ld(1, (b"\x08" + _vi(nid) if nid else b"") + _ld(2, name.encode()))
        for name, full, key in nd["attrs"]:
            body += _ld(2, _ld(1, name.encode()) + _ld(2, full.encode()) + _ld(3, key.encode()))
        for orig, slot, var in nd["slots"]:
            body += _ld(3, b"\x08" + _vi(orig) + _ld(2, slot.encode()) + b"\x18" + _vi(var))
        body += _ld(5, b"\x08\x01")                                    # has_checkpoint_values { value: true }
        graph += _ld(1, body)
    return tensors, graph


def test_reader_on_a_hand_assembled_v2_checkpoint(tmp_path):
    """PARITY UNPINNED (no TF-written file exists offline).  The reader against an index that write_bundle did not
    produce: snappy-compressed data blocks, restart interval 16 over 40-entry blocks, two data shards, a DT_STRING
    object graph, int64 and Adam-slot entries — and variable keys that do NOT have the expected text, so load_v2 can
    only find the dense layers by walking the object graph along g_net -> all_layers -> i -> 0 -> kernel / bias."""
    g = BaseFullyConnectedNet(3, 5, "g_net", [24, 16, 24], rng=np.random.RandomState(0))
    e = BaseFullyConnectedNet(5, 3, "e_net", [8, 8], rng=np.random.RandomState(1))
    tensors, graph = _keras_like_checkpoint(g, e)
    tensors["_CHECKPOINTABLE_OBJECT_GRAPH"] = graph
    prefix = str(tmp_path / "ckpt-3")
    _hand_bundle(prefix, tensors, shards=2)
    r = ck.BundleReader(prefix)
    assert r.num_shards == 2 and len(r.entries) == len(tensors) and len(tensors) > 40
    assert {r.entries[k]["shard_id"] for k in r.entries} == {0, 1}
    assert r.get_strings("_CHECKPOINTABLE_OBJECT_GRAPH", verify=True) == [graph]
    assert int(r.get_tensor("save_counter/.ATTRIBUTES/VARIABLE_VALUE")) == 3
    nodes = r.object_graph()
    key = ck.resolve_path(nodes, ["g_net", "all_layers", "2", "0", "kernel"])
    assert key == "g_net/layer_with_weights-2/kernel/.ATTRIBUTES/VARIABLE_VALUE"
    assert ck.resolve_path(nodes, ["g_net", "layer_with_weights-2", "bias"]).endswith("-2/bias/.ATTRIBUTES/VARIABLE_VALUE")
    opt = nodes[nodes[0]["children"]["g_e_optimizer"]]
    assert len(opt["slots"]) == 2 * 2 * (4 + 3) and opt["slots"][0][1] == "m"
    g2 = BaseFullyConnectedNet(3, 5, "g_net", [24, 16, 24])
    e2 = BaseFullyConnectedNet(5, 3, "e_net", [8, 8])
    ck.load_v2(prefix, g2, e2)
    assert all(np.array_equal(a, b) for a, b in zip(g.get_weights(), g2.get_weights()))
    assert all(np.array_equal(a, b) for a, b in zip(e.get_weights(), e2.get_weights()))
    with pytest.raises(KeyError):                                       # one dense layer more than the file holds
        ck.load_v2(prefix, BaseFullyConnectedNet(3, 5, "g_net", [24, 16, 24, 24]), e2)
    # a flipped byte in a data shard is caught by the per-tensor crc32c (and passes unnoticed only on request)
    for shard in (0, 1):
        path = "%s.data-%05d-of-00002" % (prefix, shard)
        raw = bytearray(open(path, "rb").read())
        raw[len(raw) // 2] ^= 0x10
        open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        for k in sorted(tensors):
            if k != "_CHECKPOINTABLE_OBJECT_GRAPH":
                r.get_tensor(k)
    # a flipped byte in a compressed index block is caught by the block checksum
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[100] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        ck.BundleReader(prefix)


def test_reader_on_a_hand_assembled_v1_checkpoint(tmp_path):
    """The tf.train.Saver layout (main_density_est.py:123,389-395) assembled by the independent code above: one shard,
    uncompressed 7-entry blocks, Adam slots and beta powers beside the variables; load_v1 with its default
    verify=True, and a corrupted data file is refused."""
    g = Generator(2, 3, "g_net", 3, 16, rng=np.random.RandomState(2))
    h = Generator(3, 2, "h_net", 3, 8, rng=np.random.RandomState(3))
    tensors = {"beta1_power": np.array(0.5, np.float32), "beta2_power": np.array(0.9, np.float32)}
    for net in (g, h):
        for (wn, bn), w, b in zip(ck.v1_names(net.name, 4), net.weights, net.biases):
            for name, arr in ((wn, w), (bn, b)):
                tensors[name] = arr
                tensors[name + "/Adam"] = np.zeros_like(arr)
                tensors[name + "/Adam_1"] = np.ones_like(arr)
    prefix = str(tmp_path / "model.ckpt-best")
    _hand_bundle(prefix, tensors, shards=1, per_block=7, compress=False)
    g2, h2 = Generator(2, 3, "g_net", 3, 16), Generator(3, 2, "h_net", 3, 8)
    ck.load_v1(prefix, g2, h2)
    assert all(np.array_equal(a, b) for a, b in zip(g.get_weights(), g2.get_weights()))
    assert all(np.array_equal(a, b) for a, b in zip(h.get_weights(), h2.get_weights()))
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    off = ck.BundleReader(prefix).entries["g_net/fully_connected_2/weights"]["offset"]
    raw[off + 9] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        ck.load_v1(prefix, g2, h2)
    ck.load_v1(prefix, g2, h2, verify=False)                            # on request the damage passes unnoticed
    assert not np.array_equal(g2.weights[2], g.weights[2])


def test_own_writer_round_trips_object_graph_and_shards(tmp_path):
    g = BaseFullyConnectedNet(2, 2, "g_net", [16, 16], rng=np.random.RandomState(0))
    e = BaseFullyConnectedNet(2, 2, "e_net", [8, 8], rng=np.random.RandomState(1))
    prefix = str(tmp_path / "ckpt-2")
    ck.save_v2(prefix, g, e, shards=3)
    assert all(os.path.exists("%s.data-%05d-of-00003" % (prefix, k)) for k in range(3))
    r = ck.BundleReader(prefix)
    nodes = r.object_graph()
    assert ck.resolve_path(nodes, ["e_net", "all_layers", "1", "0", "bias"]) == \
        "e_net/all_layers/1/0/bias/.ATTRIBUTES/VARIABLE_VALUE"
    assert r.get_strings(ck.OBJECT_GRAPH_KEY, verify=True)[0] == ck._encode_object_graph(
        ck.v2_object_graph([("g_net", g), ("e_net", e)]))
    g2, e2 = BaseFullyConnectedNet(2, 2, "g_net", [16, 16]), BaseFullyConnectedNet(2, 2, "e_net", [8, 8])
    ck.load_v2(prefix, g2, e2)
    assert all(np.array_equal(a, b) for a, b in zip(g.get_weights(), g2.get_weights()))
    assert all(np.array_equal(a, b) for a, b in zip(e.get_weights(), e2.get_weights()))


def test_native_crc32c_equals_the_python_one():
    data = np.random.RandomState(0).bytes(100003)
    assert ck.crc32c(data) == ck._crc32c_py(data)
    assert ck.crc32c(data[5000:], ck.crc32c(data[:5000])) == ck._crc32c_py(data)
    assert _crc32c_bitwise(data[:3000]) == ck._crc32c_py(data[:3000])
    lib = _lib.load()
    assert lib.rt_crc32c(b"123456789", 9, 0) == 0xE3069283


def test_latest_checkpoint(tmp_path):
    assert ck.latest_checkpoint(str(tmp_path)) is None
    ck.write_checkpoint_state(str(tmp_path), "ckpt-9")
    assert ck.latest_checkpoint(str(tmp_path)) == os.path.join(str(tmp_path), "ckpt-9")


# ---- sharding ---------------------------------------------------------------------------------------------
def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 9, 10000, 1000003):
        for ws in (1, 2, 3, 8):
            blocks = [rdist.shard_bounds(n, ws, r) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            per = -(-n // ws) if n else 0
            assert all(hi - lo <= per for lo, hi in blocks)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n", [11, 2])
def test_sharded_map_world_size_2_gloo(tmp_path, n):
    """N > 1 path on CPU: two gloo ranks each evaluate their contiguous block with the global index base
    and the all-gather assembles exactly the unsharded answer."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np
        import torch.distributed as dist
        from roundtrip_amd import dist as rdist
        dist.init_process_group("gloo")
        n = %d
        pts = np.arange(n * 3, dtype=np.float64).reshape(n, 3)
        calls = []
        def fn(block, base):
            calls.append((base, len(block)))
            return block.sum(axis=1) * 1000.0 + base + np.arange(len(block))      # depends on the GLOBAL index
        out = rdist.sharded_map(pts, fn)
        want = pts.sum(axis=1) * 1000.0 + np.arange(n)
        lo, hi = rdist.shard_bounds(n, 2, dist.get_rank())
        assert calls == ([(lo, hi - lo)] if hi > lo else []), calls
        assert out.dtype == np.float64 and np.array_equal(out, want), (out, want)
        dist.barrier(); dist.destroy_process_group()
        print("rank ok")
    """ % (ROOT, n)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("rank ok") == 2


def test_weighted_shard_bounds_partition_the_points():
    """shard_bounds with rank weights: contiguous, disjoint, covering, monotone in the weights; equal weights land within
    one point of the equal cut; bad weights are refused."""
    from roundtrip_amd import dist as rdist
    rng = np.random.RandomState(0)
    for n in (0, 1, 2, 7, 100, 1000003):
        for ws in (1, 2, 3, 8):
            w = rng.uniform(0.5, 2.0, ws)
            cuts = [rdist.shard_bounds(n, ws, r, w) for r in range(ws)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(ws - 1)) and all(lo <= hi for lo, hi in cuts)
            for r, (lo, hi) in enumerate(cuts):
                assert abs((hi - lo) - n * w[r] / w.sum()) <= 1.0 + 1e-9
            eq = [rdist.shard_bounds(n, ws, r, [1.0] * ws) for r in range(ws)]
            assert all(abs((hi - lo) - n / ws) <= 1.0 for lo, hi in eq)
    with pytest.raises(ValueError):
        rdist.shard_bounds(10, 2, 0, [1.0])
    with pytest.raises(ValueError):
        rdist.shard_bounds(10, 2, 0, [1.0, 0.0])


@pytest.mark.parametrize("n", [11, 3, 1])
def test_weighted_sharded_map_world_size_2_gloo(tmp_path, n):
    """Ranks of unequal speed (set_rank_weights): the blocks are cut 3 : 1, every rank contributes the LARGEST block's
    length to the all-gather (padded) and the result is the unsharded answer, bit for bit; back to equal blocks
    afterwards."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np
        import torch.distributed as dist
        from roundtrip_amd import dist as rdist
        dist.init_process_group("gloo")
        n = %d
        pts = np.arange(n * 3, dtype=np.float64).reshape(n, 3)
        want = pts.sum(axis=1) * 1000.0 + np.arange(n)
        for weights in ([3.0, 1.0], [1.0, 3.0], None):
            rdist.set_rank_weights(weights)
            calls = []
            def fn(block, base):
                calls.append((base, len(block)))
                return block.sum(axis=1) * 1000.0 + base + np.arange(len(block))      # depends on the GLOBAL index
            out = rdist.sharded_map(pts, fn)
            lo, hi = rdist.shard_bounds(n, 2, dist.get_rank(), weights)
            assert calls == ([(lo, hi - lo)] if hi > lo else []), (weights, calls)
            assert out.dtype == np.float64 and np.array_equal(out, want), (weights, out, want)
        if n == 11:
            assert rdist.shard_bounds(n, 2, 0, [3.0, 1.0]) == (0, 8) and rdist.shard_bounds(n, 2, 1, [3.0, 1.0]) == (8, 11)
        dist.barrier(); dist.destroy_process_group()
        print("rank ok")
    """ % (ROOT, n)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("rank ok") == 2


def test_rendezvous_on_a_port_of_its_own(tmp_path):
    """RT_COMM_PORT: the id exchange uses exactly that port (bench.py --gpus N picks a free one for its ranks), not
    MASTER_PORT + 1 .. + 16 — which a neighbouring job's MASTER_PORT may be."""
    port, comm_port = _free_port(), _free_port()
    script = tmp_path / "r.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        from roundtrip_amd import dist as rdist
        rank, ws, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        uid = rdist.exchange_id(lambda: bytes([(3 * i + rank) %% 256 for i in range(128)]), rank, ws, "127.0.0.1", port,
                                timeout=60.0)
        assert uid == bytes([(3 * i) %% 256 for i in range(128)])
        print("rank ok")
    """ % (ROOT,)))
    # the whole MASTER_PORT + 1 .. + 16 window is occupied by foreign listeners: only RT_COMM_PORT can work
    blockers = []
    for k in range(1, 17):
        b = socket.socket()
        try:
            b.bind(("127.0.0.1", port + k))
            b.listen(1)
            blockers.append(b)
        except OSError:
            b.close()
    try:
        procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port)], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, env=dict(os.environ, RT_COMM_PORT=str(comm_port)))
                 for r in range(2)]
        outs = [p.communicate(timeout=120) for p in procs]
    finally:
        for b in blockers:
            b.close()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "rank ok" in so, so + se


def test_all_gather_ranges_world_size_2_gloo(tmp_path):
    """The second sharding axis on CPU: two gloo ranks each hold their range of a query point's pairs (axis 1); the
    gather returns the whole array on both — for a total that divides evenly, one that does not, and one smaller than
    the world (rank 1 holds nothing)."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np
        import torch.distributed as dist
        from roundtrip_amd import dist as rdist
        dist.init_process_group("gloo")
        for total in (8, 7, 1):
            whole = np.random.RandomState(total).standard_normal((3, total, 2, 2))
            lo, hi = rdist.shard_bounds(total, 2, dist.get_rank())
            got = rdist.all_gather_ranges(whole[:, lo:hi], total, axis=1)
            assert got.shape == whole.shape and np.array_equal(got, whole), (total, got.shape)
        dist.barrier(); dist.destroy_process_group()
        print("rank ok")
    """ % (ROOT,)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("rank ok") == 2


def test_rendezvous_hands_rank0s_id_to_every_rank(tmp_path):
    """The TCP exchange of RCCL's unique id (roundtrip_amd.dist.exchange_id), four processes on CPU: every rank ends
    with rank 0's 128 bytes, although a foreign server (one that answers garbage) sits on the first candidate port and
    rank 0 starts late.  A second round (epoch 1) in the same processes works too."""
    port = _free_port()
    foreign = socket.socket()
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        foreign.bind(("127.0.0.1", port + 1))
        foreign.listen(8)
    except OSError:
        foreign = None                              # taken by someone else: equally foreign
    script = tmp_path / "r.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        from roundtrip_amd import dist as rdist
        rank, ws, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        if rank == 0:
            time.sleep(1.0)
        for epoch in (0, 1):
            uid = rdist.exchange_id(lambda: bytes([(7 * i + rank + epoch) %% 256 for i in range(128)]), rank, ws,
                                    "127.0.0.1", port, timeout=60.0, epoch=epoch)
            assert uid == bytes([(7 * i + epoch) %% 256 for i in range(128)]), (rank, epoch)
        print("rank ok")
    """ % (ROOT,)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "4", str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(4)]
    try:
        if foreign is not None:                     # answer the first peers that knock with something that is not ours
            foreign.settimeout(0.5)
            t_end = __import__("time").time() + 3.0
            while __import__("time").time() < t_end:
                try:
                    c, _ = foreign.accept()
                    c.sendall(b"HTTP/1.1 400 nope")
                    c.close()
                except socket.timeout:
                    pass
        outs = [p.communicate(timeout=120) for p in procs]
    finally:
        if foreign is not None:
            foreign.close()
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "rank ok" in o, o + e


def test_rendezvous_tells_the_peers_when_rank0_cannot_make_the_id(tmp_path):
    """If rank 0's make_id raises (RCCL missing, say), every peer gets the error text and fails within seconds instead of
    waiting out its timeout, and rank 0 re-raises its own exception."""
    import time
    port = _free_port()
    script = tmp_path / "r.py"
    script.write_text(textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        from roundtrip_amd import dist as rdist
        rank, ws, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        def make_id():
            raise OSError("cannot load RCCL (test)")
        t0 = time.time()
        try:
            rdist.exchange_id(make_id, rank, ws, "127.0.0.1", port, timeout=120.0)
            print("no error")
        except OSError as e:
            print("rank0 error:", e, "after %%.0f s" %% (time.time() - t0))
        except RuntimeError as e:
            print("peer error:", e, "after %%.0f s" %% (time.time() - t0))
    """ % (ROOT,)))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3", str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(3)]
    try:
        outs = [p.communicate(timeout=100) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert time.time() - t0 < 60
    assert "rank0 error: cannot load RCCL (test)" in outs[0][0], outs[0]
    for o, e in outs[1:]:
        assert "peer error:" in o and "OSError: cannot load RCCL (test)" in o, o + e


def test_rccl_is_loaded_on_first_use_and_fails_loudly_without_a_gpu():
    """rt_comm_* dlopen librccl.so.1 on first use and resolve their entry points from that handle: on a machine
    without a GPU that much works, and RCCL's own refusal comes back as RT_ECOMM with its text — never a silent
    fallback.  (With a GPU the same call is exercised by the -m gpu one-rank communicator test.)"""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present: covered by test_rccl_branch_world_size_1")
    script = ("import ctypes as C, sys; sys.path.insert(0, %r); from roundtrip_amd import _lib; lib = _lib.load(); "
              "b = C.create_string_buffer(_lib.RT_COMM_ID_BYTES); rc = lib.rt_comm_unique_id(b); "
              "print('RC', rc, lib.rt_last_error(None).decode()); print('NULL', lib.rt_comm_unique_id(None))" % ROOT)
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "RC %d " % _lib.RT_ECOMM in res.stdout and "GetUniqueId" in res.stdout, res.stdout
    assert "cannot load RCCL" not in res.stdout and "NULL 1" in res.stdout


def test_bench_parent_starts_ranks_before_touching_the_gpu():
    """bench.py --gpus N without a launcher: the parent spawns N rank processes with the launcher's environment and
    relays the first failure's exit code.  Without a GPU the ranks fail at rt_ctx_create — loudly, not silently."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present: covered by the -m gpu bench test")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--grid", "4", "--share-gpu"], capture_output=True, text=True, timeout=240)
    assert res.returncode != 0
    assert "rt_ctx_create" in res.stderr and "torch" not in res.stderr.lower().replace("torch.distributed.run", "")


def test_reference_draws_mode_host_logic(tmp_path, monkeypatch):
    """is_sampler = "numpy" on CPU, with a stub in place of the GPU engine: the samples handed to the given-samples
    entry point are exactly what the oracle's statement-by-statement restatement draws from numpy's GLOBAL generator
    (seed 0; per point chisquare then multivariate_normal), a rank that starts at index 3 skips the first three points'
    draws, and the global generator is left alone."""
    from oracle import is_ref
    from roundtrip_amd import Generator, RoundtripModel
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(3)
    g = Generator(2, 3, "g_net", nb_layers=2, nb_units=16, rng=rng)
    h = Generator(3, 2, "h_net", nb_layers=2, nb_units=16, rng=rng)
    m = RoundtripModel(g, h, data="t", df=3)
    y = np.random.RandomState(4).standard_normal((6, 3))
    S = 50
    og, oh = is_ref.DenseNet(g.weights, g.biases, g.alpha), is_ref.DenseNet(h.weights, h.biases, h.alpha)   # numpy MLPs
    want, xs_ref, centres_ref = is_ref.estimate_py_with_IS(og, oh, y, sd_y=0.4, scale=0.2, sample_size=S, df=3,
                                                            return_samples=True)

    class StubEngine(object):
        device = 0
        seen = []

        def mlp_forward(self, which, x):
            assert which == 1
            return is_ref.predict(oh, x.astype(np.float64), 2).astype(np.float32)

        def is_estimate_given_samples(self, pts, centres, xs, sd, scale, df, log=True):
            StubEngine.seen.append(np.array(xs))
            keep = np.random.get_state()      # the oracle reseeds the global generator (as the reference does); the stub hides that
            out = is_ref.estimate_py_with_IS(og, oh, pts, sd_y=sd, scale=scale, sample_size=xs.shape[1], df=df, log=log,
                                             x_samples=xs)
            np.random.set_state(keep)
            return out

    m._engine = StubEngine()
    m._uploaded = {0: (id(g), g.version), 1: (id(h), h.version)}
    m.is_sampler = "numpy"
    np.random.seed(77)
    state = np.random.get_state()
    got = m.estimate_py_with_IS(y, 0, sd_y=0.4, scale=0.2, sample_size=S, save=False)
    assert np.array_equal(np.random.get_state()[1], state[1])
    assert np.array_equal(np.concatenate(StubEngine.seen), xs_ref)
    np.testing.assert_allclose(got, want, rtol=1e-12)
    StubEngine.seen = []
    tail = m._reference_draws_block(y, 0.4, 0.2, S, 3, True)(y[3:], 3)
    assert np.array_equal(np.concatenate(StubEngine.seen), xs_ref[3:])
    np.testing.assert_allclose(tail, want[3:], rtol=1e-12)


def test_every_entry_point_has_its_row_in_integration_md():
    """INTEGRATION.md's table is the maintainer's map from the reference to the C ABI: every function include/rt_is.h
    declares is named there (grouped spellings such as `rt_x[_dev]` and `rt_ctx_create/destroy` count)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "rt_is.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 36
    missing = []
    for n in names:
        spellings = [n]
        if n.endswith("_dev"):
            spellings.append(n[:-4] + "[_dev]")
        if n == "rt_ctx_destroy":
            spellings.append("rt_ctx_create/destroy")
        if not any(s in doc for s in spellings):
            missing.append(n)
    assert not missing, missing


def test_python_blocks_of_the_docs_compile():
    """The binding stubs of INTEGRATION.md and the usage block of README.md are at least syntactically Python."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 0
    for doc in ("INTEGRATION.md", "README.md"):
        text = open(os.path.join(root, doc)).read()
        for i, block in enumerate(re.findall(r"```python\n(.*?)```", text, re.S)):
            compile(block, "%s block %d" % (doc, i), "exec")
            n += 1
    assert n >= 4
