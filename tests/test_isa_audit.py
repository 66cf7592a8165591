"""Build check on the compiled ISA (no GPU): no kernel of the library reaches an s_barrier with its own LDS writes still in flight.
Round 5's root cause of the cross-stream irreproducibility was exactly that: hipcc dropped the lgkmcnt(0) of
__syncthreads() at the loop header of fa2::bwd_dkv_kernel.  tools/barrier_audit.py walks every kernel's instructions along fall-through
and branch edges; the control arm rebuilds the round-4 form of that kernel and must be flagged."""
import importlib.util
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _audit_module():
    spec = importlib.util.spec_from_file_location("barrier_audit", os.path.join(ROOT, "tools", "barrier_audit.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _compile(m, src, extra=()):
    out = os.path.join(tempfile.mkdtemp(), os.path.basename(src)[:-4] + ".s")
    subprocess.check_call([m.HIPCC] + m.FLAGS + list(extra) + ["-o", out, src], stderr=subprocess.DEVNULL)
    return out


def test_no_barrier_is_reached_with_lds_writes_in_flight():
    m = _audit_module()
    import glob
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(ROOT, "speecht5_amd", "csrc", "*.hip")))
    with ThreadPoolExecutor(8) as ex:
        files = list(ex.map(lambda s: _compile(m, s), srcs))
    flagged = [(os.path.basename(f), name) for f in files for name, _, _ in m.audit(f)]
    assert not flagged, flagged


def test_the_audit_sees_the_round4_form_of_the_dkv_kernel():
    m = _audit_module()
    f = _compile(m, os.path.join(ROOT, "speecht5_amd", "csrc", "flash_attn2.hip"), ["-DFA2_NO_LGKM_BARRIER"])
    names = [name for name, _, _ in m.audit(f)]
    assert any("bwd_dkv_kernel" in n for n in names), names


def test_gelu_poly_error_bound():
    """common.h gelu_poly (the transcendental-free GELU value of the bf16 epilogues): its fp32 instruction sequence restated in numpy
    with the constants parsed from the source, against the exact erf-GELU: |Phi error| <= 1.6e-5, |gelu error| <= 2.5e-5 on [-4, 4]
    and <= 1.6e-5 |x| beyond -- two orders below the bf16 rounding of the value it feeds."""
    import re
    import numpy as np
    from scipy.special import erf
    src = open(os.path.join(ROOT, "speecht5_amd", "csrc", "common.h")).read()
    body = src[src.index("float gelu_poly(float x)"):]
    body = body[:body.index("return x * fmaf(xc, g, 0.5f)")]
    clamp = np.float32(re.search(r"fminf\(fmaxf\(x, -(\d\.\d+)f\)", body).group(1))
    co = [np.float32(c) for c in re.findall(r"(-?\d\.\d+e[+-]\d+)f", body)]
    assert len(co) == 10 and abs(float(clamp) - 3 * 2 ** 0.5) < 1e-6
    x = np.linspace(-8, 8, 1600001).astype(np.float32)
    xc = np.clip(x, -clamp, clamp)
    v = (xc * xc).astype(np.float32)
    g = (v.astype(np.float64) * co[0] + co[1]).astype(np.float32)
    for a in co[2:]:
        g = (g.astype(np.float64) * v + a).astype(np.float32)      # (fma: one rounding)
    cdf = (xc.astype(np.float64) * g + 0.5).astype(np.float32)
    got = (x * cdf).astype(np.float32).astype(np.float64)
    xd = x.astype(np.float64)
    phi = 0.5 * (1 + erf(xd / np.sqrt(2)))
    err = np.abs(got - xd * phi)
    assert np.abs(cdf.astype(np.float64) - phi).max() <= 1.6e-5
    assert err[np.abs(x) <= 4].max() <= 2.5e-5, err[np.abs(x) <= 4].max()
    assert (err <= 2.5e-5 + 1.6e-5 * np.abs(xd)).all()


def test_conv0_gelu_table_error_bound():
    """conv0.hip gelu_tab (the matrix-core forward of conv layer 0): gelu(z) = max(z, 0) - |z| Q(|z|) with the upper tail Q = 1 - Phi
    as 255 chords on [0, 4.25], an entry (Q, rise) two floats -- restated in numpy (fp32 table, the chord and the product each one fused
    multiply-add) with the constants parsed from the source, against the exact erf-GELU: |Q error| <= 8.5e-6 inside the table,
    |gelu error| <= 1.1e-5 |z| everywhere (beyond 4.25 the last node's 1.07e-5 stands in for a tail that has vanished), <= 2e-5 in
    absolute terms for |z| <= 4 -- below the polynomial's 2.5e-5."""
    import re
    import numpy as np
    from scipy.special import erfc
    src = open(os.path.join(ROOT, "speecht5_amd", "csrc", "conv0.hip")).read()
    gmax = float(re.search(r"C0_GT_MAX = (\d+\.\d+)f", src).group(1))
    assert "C0_GT_SCALE = 255.f / C0_GT_MAX" in src and "fminf(fabsf(z) * C0_GT_SCALE, 255.f)" in src
    assert "make_float2(v0, v1 - v0)" in src and "fmaf(__builtin_amdgcn_fractf(a), e.y, e.x)" in src
    h = np.float32(gmax) / np.float32(255)
    x0 = (np.arange(256, dtype=np.float32) * h).astype(np.float32)
    v0 = (0.5 * erfc(x0.astype(np.float64) / np.sqrt(2))).astype(np.float32)
    v1 = (0.5 * erfc((x0 + h).astype(np.float64) / np.sqrt(2))).astype(np.float32)
    v1[255] = v0[255]
    rise = (v1 - v0).astype(np.float32)
    z = np.linspace(-8, 8, 1600001).astype(np.float32)
    a = np.minimum(np.abs(z) * np.float32(255.0 / gmax), np.float32(255)).astype(np.float32)
    i = a.astype(np.int32)
    fr = (a - np.floor(a)).astype(np.float32)
    q = (fr.astype(np.float64) * rise[i].astype(np.float64) + v0[i].astype(np.float64)).astype(np.float32)
    got = (-np.abs(z).astype(np.float64) * q + np.maximum(z, 0)).astype(np.float32).astype(np.float64)
    zd = z.astype(np.float64)
    Q = 0.5 * erfc(np.abs(zd) / np.sqrt(2))
    ref = np.maximum(zd, 0) - np.abs(zd) * Q            # == z Phi(z)
    err = np.abs(got - ref)
    inside = np.abs(zd) <= gmax
    assert np.abs(q.astype(np.float64) - Q)[inside].max() <= 8.5e-6, np.abs(q.astype(np.float64) - Q)[inside].max()
    assert (err <= 1.1e-5 * np.abs(zd) + 1e-7).all(), float((err - 1.1e-5 * np.abs(zd)).max())
    assert err[np.abs(zd) <= 4].max() <= 2e-5, err[np.abs(zd) <= 4].max()


def test_conv0_gelu_grad_table_error_bound():
    """conv0.hip gelu_grad_tab (the matrix-core backward of conv layer 0): gelu'(z) = Phi(z) + z phi(z) as 511 chords over [-4.35, 4.35],
    an entry (value, rise) two floats, index = clamp(z scale + 255.5, 0, 511) -- restated in numpy with the constants parsed from the
    source, against the exact derivative: |error| <= 3e-5 inside the table, <= 1.4e-4 beyond it (the end values stand in for 1 and 0) --
    the polynomial it replaces: 1.2e-4 everywhere."""
    import re
    import numpy as np
    from scipy.special import erfc
    src = open(os.path.join(ROOT, "speecht5_amd", "csrc", "conv0.hip")).read()
    R = float(re.search(r"C0_GG_R = (\d+\.\d+)f", src).group(1))
    assert "C0_GG_SCALE = 511.f / (2.f * C0_GG_R), C0_GG_OFF = 255.5f" in src
    assert "__builtin_amdgcn_fmed3f(fmaf(z, C0_GG_SCALE, C0_GG_OFF), 0.f, 511.f)" in src

    def exact(x):
        x = np.asarray(x, np.float64)
        return 0.5 * erfc(-x / np.sqrt(2)) + x * np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
    h = np.float32(2 * R) / np.float32(511)
    nodes = (np.float32(-R) + np.arange(513, dtype=np.float32) * h).astype(np.float32)
    v = exact(nodes).astype(np.float32)
    val, rise = v[:512].copy(), (v[1:513] - v[:512]).astype(np.float32)
    rise[511] = 0
    z = np.linspace(-8, 8, 1600001).astype(np.float32)
    a = np.clip((z.astype(np.float64) * np.float32(511.0 / (2 * R)) + 255.5).astype(np.float32), 0, 511).astype(np.float32)
    i = a.astype(np.int32)
    fr = (a - np.floor(a)).astype(np.float32)
    got = (fr.astype(np.float64) * rise[i] + val[i]).astype(np.float32).astype(np.float64)
    err = np.abs(got - exact(z))
    inside = np.abs(z) <= R
    assert err[inside].max() <= 3e-5, err[inside].max()
    assert err.max() <= 1.4e-4, err.max()


def test_gelu_grad_poly_error_bound():
    """common.h gelu_grad_poly (derivative of GELU in the bf16 backward epilogues), restated in numpy from the parsed constants:
    |error| <= 1.3e-4 everywhere (the exact derivative lies in [-0.13, 1.13]; the factor multiplies bf16 operands)."""
    import re
    import numpy as np
    from scipy.special import erf
    src = open(os.path.join(ROOT, "speecht5_amd", "csrc", "common.h")).read()
    body = src[src.index("float gelu_grad_poly(float x)"):]
    body = body[:body.index("return fmaf(xc, h, 0.5f)")]
    clamp = np.float32(re.search(r"fminf\(fmaxf\(x, -(\d\.\d+)f\)", body).group(1))
    co = [np.float32(c) for c in re.findall(r"(-?\d\.\d+e[+-]\d+)f", body)]
    assert len(co) == 11
    x = np.linspace(-10, 10, 2000001).astype(np.float32)
    xc = np.clip(x, -clamp, clamp)
    v = (xc * xc).astype(np.float32)
    h = (v.astype(np.float64) * co[0] + co[1]).astype(np.float32)
    for a in co[2:]:
        h = (h.astype(np.float64) * v + a).astype(np.float32)
    got = (xc.astype(np.float64) * h + 0.5).astype(np.float32).astype(np.float64)
    xd = x.astype(np.float64)
    exact = 0.5 * (1 + erf(xd / np.sqrt(2))) + xd * np.exp(-xd * xd / 2) / np.sqrt(2 * np.pi)
    assert np.abs(got - exact).max() <= 1.3e-4, np.abs(got - exact).max()
