"""Times the row-plan builder alone and checks its invariants (see tools/row_plan_probe.hip for the build line).

Argument: an .npy / .npz of detected-human counts [steps, 4096] from a real rollout (tools/det_counts_sample.npz); without one, random
counts.  Prints per phase of group 0: row counts + offsets + histogram, counting sort, the class loop (capacities + scan / hand-out), total."""
import ctypes as C, numpy as np, torch, os, sys
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "row_plan_probe.so"))
det_all = np.load(sys.argv[1]) if len(sys.argv) > 1 else None
if det_all is not None and hasattr(det_all, "files"):
    det_all = det_all[det_all.files[0]]   # .npz: tools/det_counts_sample.npz holds one array
E, H = 4096, 20
if det_all is None:
    rs = np.random.RandomState(0); det_all = rs.randint(1, 13, size=(4, E))
L.rp_test.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
words = L.rp_test_words(E)
plan = torch.zeros(words, dtype=torch.int32, device="cuda"); tim = torch.zeros(16, dtype=torch.int64, device="cuda")
for s in (0, 7, 11, 22):
    det = torch.from_numpy(det_all[s % len(det_all)].astype(np.float32)).cuda()
    for rep in range(3):
        tim.zero_(); L.rp_test(E, H, det.data_ptr(), plan.data_ptr(), tim.data_ptr(), None); torch.cuda.synchronize()
    t = tim.cpu().numpy()
    print("   class loop: capacities + scan %.1f us, hand-out %.1f us" % (t[8]*0.01, t[10]*0.01))
    p = plan.cpu().numpy()
    d = det_all[s % len(det_all)].astype(np.int64).clip(1, H)
    ro = p[8:8 + E + 1]
    assert (ro == np.concatenate([[0], np.cumsum(d)])).all(), "row offsets"
    NW, n, total, T = int(p[1]), int(p[2]), int(p[3]), int(p[6])
    oc = 8 + ((E + 1 + 3) & ~3); oi = oc + 1024
    cnt = p[oc:oc + T]; items = p[oi:oi + 1024 * 64].reshape(1024, 64)
    seen = np.zeros(E, int); loads = []
    for b_ in range(T):
        ids = items[b_, :cnt[b_]] & 0xffff; rows = items[b_, :cnt[b_]] >> 16
        assert (rows == d[ids]).all(); seen[ids] += 1; loads.append(int(rows.sum()))
    assert (seen == 1).all(), "every env exactly once"
    loads = np.array(loads)
    print("   tiles %d rows %d: tile rows %d..%d, envs per tile %d..%d, max blocks/WG %d" % (T, total, loads.min(), loads.max(), cnt.min(), cnt.max(), max(sum((loads[c_ + j * NW] + 15) // 16 for j in range(n)) for c_ in range(NW))))
    print("step %d valid %s: pass1 %.1f us, pass2 %.1f us, fill+handout %.1f us, tail %.1f us, total %.1f us" % (
        s, hex(p[0]), (t[1]-t[0])*0.01, (t[2]-t[1])*0.01, (t[3]-t[2])*0.01, (t[4]-t[3])*0.01, (t[4]-t[0])*0.01))
