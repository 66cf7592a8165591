"""Lists every kernel of the built library whose code object uses SCRATCH (private_segment_fixed_size > 0) or spills: no GPU needed.
Round 4 found bwd_dq_kernel<true> keeping three running sums in scratch (68 scratch loads + stores per tile, 2.2x kernel time) -- a
codegen accident no test can see.  tests/test_state_dict.py runs this as a build check."""
import os, re, subprocess, sys, tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernels(lib):
    raw = open(lib, "rb").read()
    out = []
    pos = 0
    while True:
        i = raw.find(b"\x7fELF\x02\x01\x01\x40", pos)      # ELF64, little endian, OS ABI 64 = AMDGPU HSA
        if i < 0:
            break
        # e_shoff + e_shnum * e_shentsize bounds the image
        shoff = int.from_bytes(raw[i + 0x28:i + 0x30], "little")
        shentsize = int.from_bytes(raw[i + 0x3A:i + 0x3C], "little")
        shnum = int.from_bytes(raw[i + 0x3C:i + 0x3E], "little")
        end = i + shoff + shentsize * shnum
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(raw[i:end]); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
            out.append((name.group(1), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("vgpr_count"), g("sgpr_spill_count")))
        pos = end
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "speecht5_amd", "libspeecht5_hip.so")
    ks = kernels(lib)
    bad = [k for k in ks if k[1] or k[2]]
    print(f"{len(ks)} kernels in {lib}; {len(bad)} use scratch")
    for n, priv, vsp, vg, ssp in bad:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        print(f"  scratch {priv:5d} B  vgpr spills {vsp:3d}  vgprs {vg:3d}  {d[:150]}")
