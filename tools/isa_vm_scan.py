"""Memory-operation skeleton of every kernel in a .hip source: where loads are issued and where they are waited for.

    python tools/isa_vm_scan.py go_slam_amd/csrc/neus_bwd.hip [extra hipcc flags ...]

Compiles the file for gfx950 (device code only, no GPU needed) and prints, per kernel, its vector-memory instructions
and vmcnt waits in program order:  L = load, S = store / atomic, wN = s_waitcnt vmcnt(N), [ = loop header, B = barrier.
What to look for (round 6: the patterns behind five of the changes in DESIGN section 0):
  * `L w0 L w0 L w0 ...`  -- a chain of dependent round trips: every load is consumed (converted, compared, used as an
    index) before the next one is issued.  Request first, convert later; uniform index chains belong on scalar loads
    (constant address space).
  * `L L L L w0` inside a loop right after `S` -- a wait for ALL outstanding operations, stores' acknowledgements
    included: some younger memory operation sits behind a branch, so the compiler cannot count what is in flight.
  * a prefetch whose `w0` follows its own `L` at once -- something derived from the loaded value is computed at the
    request (the registers must hold what the loads return and nothing else).
"""
import re
import subprocess
import sys
import tempfile


def scan(src, extra):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-munsafe-fp-atomics", "-Igo_slam_amd/csrc", "-Iinclude", "-S", "--cuda-device-only", src, "-o", f.name]
                       + extra, check=True, stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:\s*;? *@", l)] + [len(lines)]
    for a, b in zip(starts, starts[1:]):
        seq = []
        for l in lines[a:b]:
            if re.search(r"\b(global|buffer|flat)_load", l):
                seq.append("L")
            elif re.search(r"\b(global|buffer|flat)_(store|atomic)", l):
                seq.append("S")
            elif "s_waitcnt" in l and "vmcnt" in l:
                seq.append("w" + re.search(r"vmcnt\((\d+)\)", l).group(1))
            elif "Loop Header" in l:
                seq.append("[")
            elif "s_barrier" in l:
                seq.append("B")
        print(lines[a].split(":")[0])
        print("    " + " ".join(seq))


if __name__ == "__main__":
    scan(sys.argv[1], sys.argv[2:])
