"""ISA audit of the env kernels (no GPU needed): compiles csrc/dcc_env.hip to gfx950 assembly and reports, per kernel instantiation,
registers / occupancy / code size and what sits INSIDE its loops that can stall a streaming wave: vector loads, `s_waitcnt vmcnt(0)`
that the compiler inserted (the pacing waits written as inline asm are listed separately), and -- for the role-specialised kernels --
the same for the observation wave's loop alone.  Round 6 found an accidental per-env-step `global_load` + `vmcnt(0)` in that loop this
way (a conditional lvalue on kernel-argument members); `--check` exits non-zero if a vector load is back in any observation-wave loop.
usage: python tools/isa_audit.py [--check] [--all]        (default: the BASELINE instantiations only)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dynamic-coverage-control_amd", "csrc", "dcc_env.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
         "-I" + os.path.join(ROOT, "include"), "-DDCC_BUILDING=1", "--cuda-device-only", "-S", "-w"]


def demangle(names):
    out = subprocess.run(["/usr/bin/c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {n: d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for n, d in zip(names, out)}


def main():
    check, every = "--check" in sys.argv, "--all" in sys.argv
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "dcc_env.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", asm, SRC], check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and "dcc_" in l]
    names = demangle([n for _, n in starts])
    bad = 0
    print("%-46s %5s %4s %6s | in loops: %5s %9s %7s | observation-wave loop: %5s %9s %7s" % (
        "kernel", "VGPR", "occ", "code B", "vload", "vmcnt(0)", "paced", "vload", "vmcnt(0)", "paced"))
    for (i, n) in starts:
        name = names[n]
        if not every and not re.search(r"<(\d+, )?\d, (true|false), (8, 64|16, 256|4, 20|4, 16)>|<16, \d, true, 0, 0>", name):
            continue
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))      # (a kernel may hold several s_endpgm)
        meta = "\n".join(lines[end:end + 80])
        g = lambda k: (re.search(r"; %s: (\d+)" % k, meta) or [None, "?"])[1]
        code = (re.search(r"codeLenInByte = (\d+)", meta) or [None, "?"])[1]
        body = lines[i:end]
        # loop extents: a label that is the target of a later backward branch
        label_at = {l.split(":")[0]: j for j, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        in_loop = [False] * len(body)
        for j, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in label_at and label_at[m.group(1)] < j:
                for q in range(label_at[m.group(1)], j + 1):
                    in_loop[q] = True

        def count(lo, hi):
            vl = w0 = paced = 0
            for j in range(lo, hi):
                if not in_loop[j]:
                    continue
                l = body[j]
                if re.search(r"\b(global|buffer|flat)_load", l):
                    vl += 1
                if "s_waitcnt vmcnt(0)" in l:
                    if j > 0 and "ASMSTART" in body[j - 1]:
                        paced += 1
                    else:
                        w0 += 1
            return vl, w0, paced

        tot = count(0, len(body))
        obs = ("", "", "")
        if "roles_kernel" in name:
            # the observation wave: from its s_setprio to the header of the physics wave's step loop (the next depth-1 loop header after it)
            sp = next((j for j, l in enumerate(body) if "s_setprio" in l), None)
            heads = [j for j, l in enumerate(body) if "Loop Header: Depth=1" in l and sp is not None and j > sp]
            if sp is not None and len(heads) >= 2:
                obs = count(sp, heads[1])
                if obs[0] > 0:
                    bad += 1
        print("%-46s %5s %4s %6s |           %5d %9d %7d |                        %5s %9s %7s" % (
            name[:46], g("NumVgprs"), g("Occupancy"), code, tot[0], tot[1], tot[2], obs[0], obs[1], obs[2]))
    print("\nvload = vector loads inside loops (the physics waves' action chunks and state loads are expected; an observation-wave loop must have NONE);\n"
          "vmcnt(0) = compiler-inserted full drains inside loops; paced = the `s_waitcnt vmcnt(0)` of KParams::obs_drain (inline asm, one per flush site).")
    if check and bad:
        raise SystemExit("%d observation-wave loop(s) hold a vector load again" % bad)


if __name__ == "__main__":
    main()
