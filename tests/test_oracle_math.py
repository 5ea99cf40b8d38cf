"""The oracle's definitions of the WGSL built-ins the reference leaves to the backend (sin, cos, pow) must be
accurate: any correctly-behaving WGSL implementation is a legal reference execution, a sloppy one is not."""
import numpy as np


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.maximum(np.abs(ref32), np.float32(1e-30)))
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_sincos_accuracy(orc):
    L = orc.lib()
    x = np.concatenate([np.linspace(0, 6.2831855, 2_000_001), [0.0, np.pi / 2, np.pi, 3 * np.pi / 2]]).astype("<f4")
    s, c = np.zeros_like(x), np.zeros_like(x)
    L.orc_probe_sincos(orc._p(x), orc._p(s), orc._p(c), x.size)
    x64 = x.astype(np.float64)
    assert np.abs(s - np.sin(x64)).max() < 1.5e-7  # absolute: ~1 ulp of 1.0
    assert np.abs(c - np.cos(x64)).max() < 1.5e-7
    assert (s * s + c * c - 1).__abs__().max() < 5e-7
    # alpha = 2*pi*u with u = 0 and u = 1 (rng_next_float's range is inclusive)
    s0, c0 = np.zeros(2, "<f4"), np.zeros(2, "<f4")
    xs = np.array([0.0, np.float32(2.0) * np.float32(3.1415927)], "<f4")
    L.orc_probe_sincos(orc._p(xs), orc._p(s0), orc._p(c0), 2)
    assert s0[0] == 0.0 and c0[0] == 1.0 and abs(s0[1]) < 4e-7 and abs(c0[1] - 1) < 1e-7


def test_pow_accuracy(orc):
    L = orc.lib()
    rng = np.random.default_rng(3)
    u = np.concatenate([rng.random(1_000_000), 2.0 ** -np.arange(1, 33)]).astype("<f4")
    out = np.zeros_like(u)
    y = np.full_like(u, 0.33333)
    L.orc_probe_pow(orc._p(u), orc._p(y), orc._p(out), u.size)
    assert ulp_err(out, u.astype(np.float64) ** np.float64(np.float32(0.33333))).max() < 8  # shade.wgsl:120
    b = (2 * u).astype("<f4")
    y5 = np.full_like(u, 5.0)
    L.orc_probe_pow(orc._p(b), orc._p(y5), orc._p(out), u.size)
    ref = b.astype(np.float64) ** 5
    big = ref > 1e-30
    # shade.wgsl:161. WGSL defines pow's accuracy as that of exp2(y * log2(x)), so the error grows with
    # |y * log2 x|: a few ulp for ordinary bases, tens of ulp for bases near 2^-20
    err = ulp_err(out[big], ref[big])
    near = b[big] > 2.0 ** -6
    assert err[near].max() < 40 and err.max() < 128
    # special cases the shaders can reach
    sp = np.array([0.0, 1.0, -1e-7, 1e-30], "<f4")
    ys = np.array([0.33333, 0.33333, 5.0, 5.0], "<f4")
    o = np.zeros(4, "<f4")
    L.orc_probe_pow(orc._p(sp), orc._p(ys), orc._p(o), 4)
    assert o[0] == 0.0 and o[1] == 1.0 and np.isnan(o[2]) and o[3] == 0.0


def test_normalize_accuracy(orc):
    """normalize(v) is defined as v * (1 / length(v)) since round 4 (oracle/wfpt_oracle.c: v3_normalize; the device: normalize3).
    WGSL gives the built-in the accuracy of v / length(v), i.e. the division's 2.5 ULP on top of whatever length() returned. So
    each component must lie within 2.5 ULP of the exact quotient by the SAME fp32 length (it lies within 1.5: a correctly rounded
    reciprocal and a correctly rounded product); against the true unit vector the old and the new definition are equally good."""
    import ctypes as C
    L = orc.lib()
    L.orc_probe_normalize3.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(11)
    n = 1_000_000
    v = (rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype("<f4")
    v[:1000] = rng.normal(size=(1000, 3)).astype("<f4") * np.float32(1e-12)  # tiny vectors: the squares stay normal down to ~1e-19
    out = np.zeros_like(v)
    L.orc_probe_normalize3(orc._p(v), orc._p(out), n)
    length32 = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])  # fp32, the definition's operation order
    assert length32.dtype == np.float32
    v64 = v.astype(np.float64)
    quotient = v64 / length32.astype(np.float64)[:, None]
    sel = np.abs(quotient) > 1e-30
    err = ulp_err(out[sel], quotient[sel]).max()
    assert err <= 1.5, err
    true_unit = v64 / np.linalg.norm(v64, axis=1, keepdims=True)
    per_component = (v64 / length32.astype(np.float64)[:, None]).astype(np.float32)  # rounds 1-3: a division per component
    assert ulp_err(out[sel], true_unit[sel]).max() < ulp_err(per_component[sel], true_unit[sel]).max() + 0.5
    assert np.abs(np.linalg.norm(out.astype(np.float64), axis=1) - 1.0).max() < 3e-7
