"""Lol's protobuf ring-element messages (lol/Lol.proto: Rq, RqProduct; conventions of
lol/Crypto/Lol/Types/IZipVector.hs:127-205) — SURVEY.md §8f N3.

The reference ships no serialized samples, so the codec in liblolhip (written against the
wire specification) is pinned against google.protobuf itself: the same two messages are
declared through a dynamic descriptor, and bytes produced by either side must parse on the
other — for the unpacked encoding hprotoc emits and for the packed one.  Plus the Lol
conventions: centred lifts on write, `reduce` on read, one Rq per modulus in tuple order.
"""
import numpy as np
import pytest

from oracle import lolmath as lm
from oracle.oracle import Params


def _messages():
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="lol_wire_test.proto", package="crypto.proto.lol", syntax="proto2")
    F = descriptor_pb2.FieldDescriptorProto
    rq = fd.message_type.add(name="Rq")
    rq.field.add(name="m", number=1, type=F.TYPE_UINT32, label=F.LABEL_REQUIRED)
    rq.field.add(name="q", number=2, type=F.TYPE_UINT64, label=F.LABEL_REQUIRED)
    rq.field.add(name="xs", number=3, type=F.TYPE_SINT64, label=F.LABEL_REPEATED)
    rqp = fd.message_type.add(name="RqPacked")                      # same message, packed xs
    rqp.field.add(name="m", number=1, type=F.TYPE_UINT32, label=F.LABEL_REQUIRED)
    rqp.field.add(name="q", number=2, type=F.TYPE_UINT64, label=F.LABEL_REQUIRED)
    f = rqp.field.add(name="xs", number=3, type=F.TYPE_SINT64, label=F.LABEL_REPEATED)
    f.options.packed = True
    prod = fd.message_type.add(name="RqProduct")
    prod.field.add(name="rqlist", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".crypto.proto.lol.Rq")
    prodp = fd.message_type.add(name="RqProductPacked")
    prodp.field.add(name="rqlist", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".crypto.proto.lol.RqPacked")
    poly = fd.message_type.add(name="RqPolynomial")                # lol-apps/SHE.proto
    poly.field.add(name="coeffs", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".crypto.proto.lol.RqProduct")
    tr = fd.message_type.add(name="TypeRep")
    tr.field.add(name="a", number=1, type=F.TYPE_UINT64, label=F.LABEL_REQUIRED)
    tr.field.add(name="b", number=2, type=F.TYPE_UINT64, label=F.LABEL_REQUIRED)
    ks = fd.message_type.add(name="KSHint")
    ks.field.add(name="hint", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".crypto.proto.lol.RqPolynomial")
    ks.field.add(name="gad", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_REQUIRED, type_name=".crypto.proto.lol.TypeRep")
    rmsg = fd.message_type.add(name="R")
    rmsg.field.add(name="m", number=1, type=F.TYPE_UINT32, label=F.LABEL_REQUIRED)
    rmsg.field.add(name="xs", number=2, type=F.TYPE_SINT64, label=F.LABEL_REPEATED)
    kq = fd.message_type.add(name="Kq")
    kq.field.add(name="m", number=1, type=F.TYPE_UINT32, label=F.LABEL_REQUIRED)
    kq.field.add(name="q", number=2, type=F.TYPE_UINT64, label=F.LABEL_REQUIRED)
    kq.field.add(name="xs", number=3, type=F.TYPE_DOUBLE, label=F.LABEL_REPEATED)
    kqp = fd.message_type.add(name="KqProduct")
    kqp.field.add(name="kqlist", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".crypto.proto.lol.Kq")
    lin = fd.message_type.add(name="LinearRq")
    lin.field.add(name="e", number=1, type=F.TYPE_UINT32, label=F.LABEL_REQUIRED)
    lin.field.add(name="r", number=2, type=F.TYPE_UINT32, label=F.LABEL_REQUIRED)
    lin.field.add(name="coeffs", number=3, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".crypto.proto.lol.RqProduct")
    sk = fd.message_type.add(name="SecretKey")                      # lol-apps/SHE.proto
    sk.field.add(name="sk", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REQUIRED, type_name=".crypto.proto.lol.R")
    sk.field.add(name="v", number=2, type=F.TYPE_DOUBLE, label=F.LABEL_REQUIRED)
    th = fd.message_type.add(name="TunnelHint")
    th.field.add(name="func", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REQUIRED, type_name=".crypto.proto.lol.LinearRq")
    th.field.add(name="hint", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".crypto.proto.lol.KSHint")
    for i, nm in enumerate(("e", "r", "s")):
        th.field.add(name=nm, number=3 + i, type=F.TYPE_UINT32, label=F.LABEL_REQUIRED)
    th.field.add(name="p", number=6, type=F.TYPE_UINT64, label=F.LABEL_REQUIRED)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    cls = (lambda n: get(pool.FindMessageTypeByName("crypto.proto.lol." + n))) if get else \
          (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName("crypto.proto.lol." + n)))
    _messages.KSHint = cls("KSHint")
    for nm in ("R", "KqProduct", "LinearRq", "SecretKey", "TunnelHint"):
        setattr(_messages, nm, cls(nm))
    return cls("RqProduct"), cls("RqProductPacked")


def _lift(x, q):
    x = int(x) % q
    return x if 2 * x < q else x - q


CASES = [(8, [17]), (1024, [12289]), (21, [8191, lm.first_good_q(21, 2 ** 30)]),
         (16, [lm.first_good_q(16, 2 ** 60), 97, lm.first_good_q(16, 2 ** 61)])]


@pytest.mark.parametrize("m,qs", CASES)
def test_rqproduct_against_google_protobuf(m, qs):
    import lol_amd
    RqProduct, RqProductPacked = _messages()
    R = Params(lm.factor_pps(m), qs)
    rng = np.random.default_rng(m)
    xs = R.random(rng, 1)[0]                                    # [n][T]
    xs[0], xs[1 % R.n] = 0, np.array(qs) - 1
    xs[2 % R.n] = np.array(qs) // 2
    # ours -> protobuf
    data = lol_amd.rqproduct_write(m, qs, xs)
    msg = RqProduct()
    msg.ParseFromString(data)
    assert len(msg.rqlist) == len(qs)
    for t, rq in enumerate(msg.rqlist):
        assert rq.m == m and rq.q == qs[t]
        assert list(rq.xs) == [_lift(v, qs[t]) for v in xs[:, t]]              # centred lifts (toProto)
    assert msg.SerializeToString() == data                                      # byte-identical encoding
    # protobuf -> ours, unpacked and packed, with unreduced / negative representatives
    for cls in (RqProduct, RqProductPacked):
        msg = cls()
        for t, q in enumerate(qs):
            rq = msg.rqlist.add()
            rq.m, rq.q = m, q
            rq.xs.extend(int(v) - (q if i % 3 == 0 and v else 0) for i, v in enumerate(xs[:, t]))
        m2, qs2, xs2 = lol_amd.rqproduct_read(msg.SerializeToString())
        assert (m2, qs2) == (m, list(qs)) and np.array_equal(xs2, xs)           # `reduce` on read (fromProto)
    # round trip through ourselves, (-q, 0] inputs
    neg = np.where(xs > 0, xs - np.array(qs), 0)
    assert lol_amd.rqproduct_read(lol_amd.rqproduct_write(m, qs, neg))[2].tolist() == xs.tolist()


def test_rqproduct_rejects_malformed_input():
    import lol_amd
    good = lol_amd.rqproduct_write(8, [17, 97], np.arange(8).reshape(4, 2))
    for bad in (b"", good[:-1], good[:5], b"\x0a\x7f" + good, good + b"\x0a\x02\x08\x09"):   # truncated / m differs / no q
        with pytest.raises(lol_amd.LolHipError):
            lol_amd.rqproduct_read(bad)
    with pytest.raises(lol_amd.LolHipError):
        lol_amd.rqproduct_write(8, [1], np.zeros((4, 1)))


@pytest.mark.gpu
def test_ingest_rqproduct_to_crt_basis(gpu, cpuref):
    """serialized decoding-basis element -> slab -> l -> crt on the card = oracle of the same."""
    import lol_amd
    m = 2 ** 8 * 3
    qs = [lm.first_good_q(m, 2 ** 20), lm.first_good_q(m, 2 ** 59)]
    pps = lm.factor_pps(m)
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(7)
    dec = R.random(rng, 3)
    blobs = [lol_amd.rqproduct_write(m, qs, d) for d in dec]
    slab = np.stack([lol_amd.rqproduct_read(b)[2] for b in blobs])
    assert np.array_equal(slab, dec)
    assert np.array_equal(P.crt(P.l(slab)), cpuref.crt(R, cpuref.l(R, dec)))


def test_kshint_read():
    """SHE.proto KSHint: L polynomials of K RqProduct coefficients -> [L][K][n][T] slab."""
    import lol_amd
    _messages()
    m, qs = 16, [97, lm.first_good_q(16, 2 ** 40)]
    R = Params(lm.factor_pps(m), qs)
    rng = np.random.default_rng(3)
    Lh, K = 3, 2
    want = np.stack([np.stack([R.random(rng, 1)[0] for _ in range(K)]) for _ in range(Lh)])
    msg = _messages.KSHint()
    msg.gad.a, msg.gad.b = 123456789, 987654321
    for j in range(Lh):
        pl = msg.hint.add()
        for k in range(K):
            prod = pl.coeffs.add()
            for t, q in enumerate(qs):
                rq = prod.rqlist.add()
                rq.m, rq.q = m, q
                rq.xs.extend(_lift(v, q) for v in want[j, k, :, t])
    m2, qs2, xs = lol_amd.kshint_read(msg.SerializeToString())
    assert (m2, qs2) == (m, qs) and np.array_equal(xs, want)
    broken = msg.SerializeToString()[:-30]
    with pytest.raises(lol_amd.LolHipError):
        lol_amd.kshint_read(broken[:40])


def _fill_product(prod, m, qs, xs):
    for t, q in enumerate(qs):
        rq = prod.rqlist.add()
        rq.m, rq.q = m, q
        rq.xs.extend(_lift(v, q) for v in xs[:, t])


def test_kshint_write_is_the_inverse_and_byte_identical():
    import lol_amd
    _messages()
    m, qs = 16, [97, lm.first_good_q(16, 2 ** 40)]
    R = Params(lm.factor_pps(m), qs)
    rng = np.random.default_rng(4)
    Lh, K = 2, 3
    xs = np.stack([np.stack([R.random(rng, 1)[0] for _ in range(K)]) for _ in range(Lh)])
    data = lol_amd.kshint_write(m, qs, xs, gad=(11, 2 ** 63 + 5))
    msg = _messages.KSHint()
    msg.ParseFromString(data)
    assert (msg.gad.a, msg.gad.b) == (11, 2 ** 63 + 5) and len(msg.hint) == Lh and len(msg.hint[0].coeffs) == K
    assert msg.SerializeToString() == data
    assert np.array_equal(lol_amd.kshint_read(data)[2], xs)


def test_r_secretkey_kq_linearrq_tunnelhint_against_google_protobuf():
    """The remaining messages of Lol.proto / SHE.proto, written by google.protobuf, read by liblolhip."""
    import lol_amd
    _messages()
    rng = np.random.default_rng(9)
    # R and SecretKey: small signed integers, decoding basis
    sk = _messages.SecretKey()
    sk.sk.m, sk.v = 12, 2.5
    vals = [int(v) for v in rng.integers(-5, 6, size=4)]
    sk.sk.xs.extend(vals)
    assert lol_amd.secretkey_read(sk.SerializeToString()) [0:2] == (12, 2.5)
    assert lol_amd.secretkey_read(sk.SerializeToString())[2].tolist() == vals
    m_, xs_ = lol_amd.r_read(sk.sk.SerializeToString())
    assert m_ == 12 and xs_.tolist() == vals
    with pytest.raises(lol_amd.LolHipError):
        lol_amd.secretkey_read(sk.sk.SerializeToString())             # an R is not a SecretKey: v is required
    # KqProduct
    kqp = _messages.KqProduct()
    want = rng.standard_normal((6, 2))
    for t, q in enumerate((97, 193)):
        k = kqp.kqlist.add()
        k.m, k.q = 9, q
        k.xs.extend(float(v) for v in want[:, t])
    m_, qs_, got = lol_amd.kqproduct_read(kqp.SerializeToString())
    assert (m_, qs_) == (9, [97, 193]) and np.array_equal(got, want)
    # LinearRq: E = O_4 in R = O_16 -> S = O_24 say; coeffs are RqProducts over S
    ms, qs = 24, [97, lm.first_good_q(24, 2 ** 40)]
    S = Params(lm.factor_pps(ms), qs)
    coeffs = np.stack([S.random(rng, 1)[0] for _ in range(4)])
    lin = _messages.LinearRq()
    lin.e, lin.r = 4, 16
    for c in coeffs:
        _fill_product(lin.coeffs.add(), ms, qs, c)
    e, r, m2, qs2, xs = lol_amd.linearrq_read(lin.SerializeToString())
    assert (e, r, m2, qs2) == (4, 16, ms, qs) and np.array_equal(xs, coeffs)
    # TunnelHint around it, with two KSHints
    th = _messages.TunnelHint()
    th.func.CopyFrom(lin)
    th.e, th.r, th.s, th.p = 4, 16, 24, 2 ** 40 + 15
    hints = []
    for h in range(2):
        ks = th.hint.add()
        ks.gad.a, ks.gad.b = h, h + 1
        x = np.stack([np.stack([S.random(rng, 1)[0] for _ in range(2)]) for _ in range(3)])
        hints.append(x)
        for j in range(3):
            pl = ks.hint.add()
            for k in range(2):
                _fill_product(pl.coeffs.add(), ms, qs, x[j, k])
    d = lol_amd.tunnelhint_read(th.SerializeToString())
    assert (d["e"], d["r"], d["s"], d["p"]) == (4, 16, 24, 2 ** 40 + 15)
    assert np.array_equal(d["func"][4], coeffs) and len(d["hints"]) == 2
    for h in range(2):
        assert np.array_equal(d["hints"][h][2], hints[h])
    raw = th.SerializeToString()
    for bad in (raw[:-3], raw[:20], lin.SerializeToString()):         # truncated; a LinearRq is not a TunnelHint
        with pytest.raises(lol_amd.LolHipError):
            lol_amd.tunnelhint_read(bad)


def test_remaining_writers_and_homomprf_chains_against_google_protobuf():
    """Round 3: liblolhip WRITES R, SecretKey, LinearRq and TunnelHint and reads/writes the chain messages of
    lol-apps/HomomPRF.proto:18-26; google.protobuf parses every byte string back to the same content, and where
    the encoding is canonical (unpacked sint64, fields in tag order: what hprotoc emits) the bytes are identical."""
    import lol_amd
    _messages()
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    rng = np.random.default_rng(11)
    # R / SecretKey
    vals = [int(v) for v in rng.integers(-9, 10, size=6)] + [2 ** 40, -2 ** 40]
    r_bytes = lol_amd.r_write(20, vals)
    R = _messages.R(); R.ParseFromString(r_bytes)
    assert R.m == 20 and list(R.xs) == vals and R.SerializeToString() == r_bytes
    assert lol_amd.r_read(r_bytes)[1].tolist() == vals
    sk_bytes = lol_amd.secretkey_write(20, 1.75, vals)
    SK = _messages.SecretKey(); SK.ParseFromString(sk_bytes)
    assert SK.v == 1.75 and list(SK.sk.xs) == vals and SK.SerializeToString() == sk_bytes
    assert lol_amd.secretkey_read(sk_bytes)[0:2] == (20, 1.75)
    # LinearRq
    ms, qs = 24, [97, lm.first_good_q(24, 2 ** 40)]
    S = Params(lm.factor_pps(ms), qs)
    coeffs = np.stack([S.random(rng, 1)[0] for _ in range(4)])
    lin_bytes = lol_amd.linearrq_write(4, 16, ms, qs, coeffs)
    lin = _messages.LinearRq(); lin.ParseFromString(lin_bytes)
    assert (lin.e, lin.r, len(lin.coeffs)) == (4, 16, 4) and lin.SerializeToString() == lin_bytes
    e, r, m2, qs2, xs = lol_amd.linearrq_read(lin_bytes)
    assert (e, r, m2, qs2) == (4, 16, ms, qs) and np.array_equal(xs, coeffs)
    # TunnelHint = LinearRq + two KSHints + e, r, s, p
    hint_x = [np.stack([np.stack([S.random(rng, 1)[0] for _ in range(2)]) for _ in range(3)]) for _ in range(2)]
    hint_bytes = [lol_amd.kshint_write(ms, qs, x, gad=(h, h + 1)) for h, x in enumerate(hint_x)]
    th_bytes = lol_amd.tunnelhint_write(lin_bytes, hint_bytes, 4, 16, 24, 2 ** 40 + 15)
    th = _messages.TunnelHint(); th.ParseFromString(th_bytes)
    assert (th.e, th.r, th.s, th.p, len(th.hint)) == (4, 16, 24, 2 ** 40 + 15, 2) and th.SerializeToString() == th_bytes
    d = lol_amd.tunnelhint_read(th_bytes)
    assert np.array_equal(d["func"][4], coeffs) and all(np.array_equal(d["hints"][h][2], hint_x[h]) for h in range(2))
    # the chains of HomomPRF.proto, declared through a dynamic descriptor that imports the messages above by bytes
    fd = descriptor_pb2.FileDescriptorProto(name="homomprf_test.proto", package="crypto.proto.HomomPRF", syntax="proto2")
    F = descriptor_pb2.FieldDescriptorProto
    for nm in ("LinearFuncChain", "TunnelHintChain", "RoundHintChain"):
        msg = fd.message_type.add(name=nm)
        msg.field.add(name="elems", number=1, type=F.TYPE_BYTES, label=F.LABEL_REPEATED)      # a sub-message and bytes share wire type 2
    pool = descriptor_pool.DescriptorPool(); pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    cls = (lambda n: get(pool.FindMessageTypeByName("crypto.proto.HomomPRF." + n))) if get else \
          (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName("crypto.proto.HomomPRF." + n)))
    for nm, elems in (("LinearFuncChain", [lin_bytes, lin_bytes]), ("TunnelHintChain", [th_bytes, th_bytes, th_bytes]),
                      ("RoundHintChain", hint_bytes), ("RoundHintChain", [])):
        chain = lol_amd.chain_write(elems)
        g = cls(nm)(); g.ParseFromString(chain)
        assert list(g.elems) == elems and g.SerializeToString() == chain
        assert lol_amd.chain_read(chain) == elems
    assert lol_amd.tunnelhint_read(lol_amd.chain_read(lol_amd.chain_write([th_bytes]))[0])["p"] == 2 ** 40 + 15
    with pytest.raises(lol_amd.LolHipError):
        lol_amd.chain_read(lol_amd.chain_write([th_bytes])[:-2])          # truncated
