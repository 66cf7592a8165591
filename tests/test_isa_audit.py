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
