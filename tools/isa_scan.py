"""Static checks on the gfx950 ISA of the kernels (build container, no GPU): which instruction classes a kernel's loops interleave, and
two patterns that cost whole memory round trips at run time without showing in the source:

  * a global STORE directly behind `s_waitcnt vmcnt(0)`: `if (row < M) C[...] = acc + bias;` puts every store into its own basic block,
    and while the bias load is still pending as far as the compiler can tell, each block starts by draining the vector-memory counter --
    which also waits for the PREVIOUS store.  128 stores per lane became 128 serial write round trips (round 2: 0.45-0.69 ms of a 2.8 ms
    GEMM; the same pattern with loads was the whole story of the GRU sequence kernels),
  * many full `s_waitcnt vmcnt(0)` drains relative to the number of loads.
(Kernels with a fast unpredicated path keep the predicated one for partial tiles: its stores still show up in the last column.)

    python tools/isa_scan.py                      # table over every kernel of every .hip file
    python tools/isa_scan.py linear gemm3p_nt     # instruction-class string of the kernels of linear.hip whose name contains gemm3p_nt
                                                  #   M mfma, r/w LDS read/write, G/S global load/store, W s_waitcnt, B s_barrier, . VALU, , SALU"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crowdnav_prediction_attngraph_amd", "csrc")
FLAGS = {"policy": ["-DCN_BK3=64"]}


def assembly(name):
    out = os.path.join(tempfile.gettempdir(), "cn_isa_%s.s" % name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
           os.path.join(CSRC, name + ".hip"), "-o", out] + FLAGS.get(name, [])
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    """name -> list of instruction lines"""
    res, fn = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn = m.group(1)
            res[fn] = []
        elif fn is not None:
            t = l.strip()
            if t.startswith("s_endpgm"):
                fn = None
            elif t and not t.startswith((";", ".")) or re.match(r"\.LBB", t):
                res[fn].append(t)
    return res


def classify(t):
    if t.startswith("v_mfma"): return "M"
    if t.startswith(("ds_read", "ds_load")): return "r"
    if t.startswith(("ds_write", "ds_store")): return "w"
    if t.startswith(("global_load", "buffer_load", "flat_load")): return "G"
    if t.startswith(("global_store", "buffer_store", "flat_store")): return "S"
    if t.startswith("s_waitcnt"): return "W"
    if t.startswith("s_barrier"): return "B"
    if t.startswith("v_"): return "."
    if t.startswith(("s_cbranch", "s_branch")): return "J"
    if re.match(r"\.LBB", t): return "\n" + t + "\n"
    if t.startswith("s_"): return ","
    return ""


def table():
    print("%-10s %-64s %6s %6s %9s %9s %12s" % ("file", "kernel", "loads", "stores", "vmcnt(0)", "vmcnt(n)", "store<-drain"))
    for f in sorted(x[:-4] for x in os.listdir(CSRC) if x.endswith(".hip")):
        for k, ins in kernels(assembly(f)).items():
            ld = sum(i.startswith(("global_load", "buffer_load", "flat_load")) for i in ins)
            st = [n for n, i in enumerate(ins) if i.startswith(("global_store", "flat_store"))]
            w0 = sum(bool(re.match(r"s_waitcnt.*vmcnt\(0\)", i)) for i in ins)
            wn = sum(bool(re.match(r"s_waitcnt.*vmcnt", i)) for i in ins) - w0
            drained = sum(any(re.match(r"s_waitcnt.*vmcnt\(0\)", i) for i in ins[max(0, n - 6):n]) for n in st)
            if ld + len(st):
                print("%-10s %-64s %6d %6d %9d %9d %12d%s" % (f, k[:64], ld, len(st), w0, wn, drained, "   <-- stores serialised?" if drained >= 8 else ""))


if __name__ == "__main__":
    if len(sys.argv) == 1:
        table()
    else:
        for k, ins in kernels(assembly(sys.argv[1])).items():
            if len(sys.argv) < 3 or sys.argv[2] in k:
                print("==", k)
                print("".join(classify(i) for i in ins))
