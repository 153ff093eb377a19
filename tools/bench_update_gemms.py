"""Stand-alone probe of the PPO update's split-precision products (run on the GPU box).

    python tools/bench_update_gemms.py                 # the six NT shapes + the three weight-gradient shapes, M = 400 000 rows
    python tools/bench_update_gemms.py --knockout      # NT q|k|v shape with parts of the kernel compiled out (needs a library built with
                                                       #   make -C crowdnav_prediction_attngraph_amd/csrc G3FLAGS=-DCN_G3_KNOCKOUT)

Every line: time per launch (5 launches between two events), algorithmic TFLOP/s, and the error against fp64 on a slice of the rows
(NT) or on the whole product (weight gradient).  Operands are uniform-random on purpose: constant fills run ~20 % faster on this part
(clock / power), see DESIGN.md section 4.  Knock-out bits (CN_G3KO): 1 = no C stores, 4 = no MFMA, 8 = no global loads in the loop."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NT_SHAPES = [(1536, 512, 0, 0), (512, 1536, 0, 0), (512, 128, 1, 0), (256, 512, 1, 0), (128, 512, 0, 1), (512, 256, 0, 1)]  # N, K, relu, gate
TN_SHAPES = [(1536, 512, 0), (512, 128, 1), (256, 512, 1)]                                                                   # N, K, gate


def timed(run, n=5):
    import torch
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def nt(M, shapes):
    import torch
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import split_bf16
    for (N, K, relu, gate) in shapes:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        g = torch.randn(M, K, device="cuda") if gate else None
        y = torch.empty(M, N, device="cuda")
        hi, lo = split_bf16(w)

        def run():
            A.check(A.lib().cn_linear_fwd(M, N, K, A.ptr(x), K, A.ptr(g) if gate else None, A.ptr(hi), A.ptr(lo), A.ptr(b), relu, A.ptr(y), N,
                                          A.stream_ptr()), "cn_linear_fwd")
        ms = timed(run)
        errs = []
        for sl in (slice(0, 700), slice(M - 300, M)):   # first rows and the (possibly partial) last tile
            xs = x[sl].double() * ((g[sl] > 0).double() if gate else 1.0)
            ref = xs @ w.double().t() + b.double()
            if relu:
                ref = ref.clamp_min(0)
            errs.append(((y[sl].double() - ref).abs().max() / ref.abs().max()).item())
        print("NT  M %d N %4d K %4d relu %d gate %d: %.3f ms %5.0f TFLOP/s  err %.1e %.1e" % (M, N, K, relu, gate, ms, 2.0 * M * N * K / ms / 1e9, *errs), flush=True)
        del x, w, y, g


def tn(M, shapes):
    import torch
    from crowdnav_prediction_attngraph_amd import _abi as A
    for (N, K, gate) in shapes:
        dy = torch.randn(M, N, device="cuda")
        x = torch.randn(M, K, device="cuda")
        g = torch.randn(M, N, device="cuda") if gate else None
        splits = A.lib().cn_linear_wgrad_splits(M, N, K)
        part = torch.empty(splits, N, K, device="cuda")
        dbp = torch.empty(splits, N, device="cuda")
        dw = torch.empty(N, K, device="cuda")
        db = torch.empty(N, device="cuda")

        def run():
            A.check(A.lib().cn_linear_wgrad(M, N, K, A.ptr(dy), N, A.ptr(g) if gate else None, A.ptr(x), K, splits, A.ptr(part), A.ptr(dbp), A.ptr(dw),
                                            A.ptr(db), A.stream_ptr()), "cn_linear_wgrad")
        ms = timed(run)
        dyg = dy.double() * ((g > 0).double() if gate else 1.0)
        ref = dyg.t() @ x.double()
        e = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
        eb = ((db.double() - dyg.sum(0)).abs().max() / dyg.sum(0).abs().max()).item()
        print("TN  M %d N %4d K %4d gate %d splits %d: %.3f ms (incl. the partial reduction) %5.0f TFLOP/s  err dW %.1e db %.1e"
              % (M, N, K, gate, splits, ms, 2.0 * M * N * K / ms / 1e9, e, eb), flush=True)
        del dy, x, g, part, dyg, ref


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--rows", type=int, default=400000)
    ap.add_argument("--knockout", action="store_true")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    if a.child is not None:              # one process per knock-out value: the library reads CN_G3KO once
        print("CN_G3KO=%s" % a.child, flush=True)
        nt(a.rows, NT_SHAPES[:1])
        return
    if a.knockout:
        for ko in ("0", "1", "4", "5", "8", "9", "12"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rows", str(a.rows), "--child", ko], env=dict(os.environ, CN_G3KO=ko),
                               capture_output=True, text=True, timeout=600)
            print(r.stdout.strip() or r.stderr[-800:], flush=True)
        return
    nt(a.rows, NT_SHAPES)
    tn(a.rows, TN_SHAPES)
    tn(a.rows - 23, TN_SHAPES[:1])       # a row count that is not a multiple of 32: the tail split


if __name__ == "__main__":
    main()
