"""Static audit of the built kernels' ISA: every s_barrier that can be reached with this wave's LDS writes still in flight.

Round 5 found the cause of the cross-stream irreproducibility: fa2::bwd_dkv_kernel published its per-tile side
array (ds_write) and went into s_barrier behind `s_waitcnt vmcnt(0)` only -- hipcc had dropped the lgkmcnt(0) of __syncthreads()'s
release fence in that loop (it keeps it in straight-line code) -- and a wave of the block on another SIMD could read the old values
when a foreign block's LDS traffic delayed the writes.  The kernels now state the wait themselves; this tool checks the compiled code:
walks each kernel's instructions along fall-through and branch edges to a fixed point (writes at the bottom of a loop reach the
barrier at its top), tracks
"a ds_write / LDS atomic was issued since the last s_waitcnt that names lgkmcnt(0)", and reports the barriers reached in that state.
No GPU needed.  usage: barrier_audit.py [file.s ...]   (default: compiles speecht5_amd/csrc/*.hip with -S into a temp dir)"""
import glob, os, re, subprocess, sys, tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-comment", "-Xclang", "-target-feature", "-Xclang",
         "-packed-fp32-ops", "-S", "--cuda-device-only"]
WRITE = re.compile(r"^\s*ds_(write|store|add|sub|min|max|and|or|xor|cmpst|wrxchg|pk_add|append|consume|inc|dec)")
WAIT0 = re.compile(r"^\s*s_waitcnt\b.*lgkmcnt\(0\)")
WAITALL = re.compile(r"^\s*s_waitcnt\s+(0|0x0)\s*$")


def audit(path):
    out = []
    name, body = None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if re.match(r"^\s*\.end_amdhsa_kernel|^\.Lfunc_end", line):
            if any("s_barrier" in x for x in body):
                # forward data flow over (fall-through + branch) edges, iterated to a fixed point: state at a label = OR over the
                # branches that target it (loop back edges carry the writes at the bottom of a loop to the barrier at its top)
                at_label, bad = {}, set()
                changed = True
                while changed:
                    changed = False
                    pend = False
                    for i, x in enumerate(body):
                        lm = re.match(r"^(\.L[\w$.]+):", x)
                        if lm:
                            pend = pend or at_label.get(lm.group(1), False)
                            continue
                        if WRITE.match(x):
                            pend = True
                        elif WAIT0.match(x) or WAITALL.match(x):
                            pend = False
                        elif re.match(r"^\s*s_barrier", x):
                            if pend:
                                bad.add(i)
                        else:
                            bm = re.match(r"^\s*s_c?branch\w*\s+(\.L[\w$.]+)", x)
                            if bm:
                                if pend and not at_label.get(bm.group(1), False):
                                    at_label[bm.group(1)] = True
                                    changed = True
                                if x.strip().startswith("s_branch"):
                                    pend = False      # nothing falls through an unconditional branch
                if bad:
                    out.append((name, len(bad), sum(1 for x in body if re.match(r"^\s*s_barrier", x))))
            name = None
            continue
        body.append(line)
    return out


def main():
    files = sys.argv[1:]
    if not files:
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        tmp = tempfile.mkdtemp()
        procs = []
        for src in sorted(glob.glob(os.path.join(here, "speecht5_amd", "csrc", "*.hip"))):
            dst = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            procs.append(subprocess.Popen([HIPCC] + FLAGS + ["-o", dst, src], stderr=subprocess.DEVNULL))
            files.append(dst)
        for p in procs:
            p.wait()
    total = 0
    for f in files:
        for name, nbad, nbar in audit(f):
            total += 1
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:170]
            print(f"{os.path.basename(f)}: {nbad} of {nbar} barriers reachable with LDS writes in flight: {dem}")
    print(f"{total} kernels flagged")
    return total


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
