"""Long-horizon soak (run by hand on an MI355X: `python tests/soak_gpu.py`; not collected by pytest): thousands of episodes of
the full-size batched simulator with a random sample of envs compared bit for bit against the scalar oracle at every step."""
import sys, numpy as np, torch, time
sys.path.insert(0, ".")
from crowdnav_prediction_attngraph_amd import _abi as A
from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
from oracle import oracle as O
from tests.test_gpu_fullsize import _scripted
def soak(kw, E, T, nsample):
    ccfg, ocfg = A.default_env_config(nenv=E, **kw), O.default_config(nenv=E, **kw)
    env = HipEnvBatch(ccfg, E, 425)
    rs = np.random.RandomState(0)
    sample = sorted(set([0, E - 1] + rs.randint(0, E, nsample).tolist()))
    oenvs = {i: O.OracleEnv(ocfg, 425 + i) for i in sample}
    obs = env.reset()
    for i, oe in oenvs.items(): oe.reset()
    idx = torch.tensor(sample, device="cuda")
    nd = 0
    t0 = time.time()
    for t in range(T):
        act = _scripted(obs["robot_node"].view(E, 7), t)
        act_h = act[idx].cpu().numpy()
        obs, rew, done, info, epr, epl = env.step(act)
        sub = {k: obs[k][idx].cpu().numpy() for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
        rew_h, done_h, info_h = rew[idx].cpu().numpy(), done[idx].cpu().numpy(), info[idx].cpu().numpy()
        for n, (i, oe) in enumerate(oenvs.items()):
            ob, r, d, inf = oe.step(act_h[n], autoreset=True)
            assert bool(done_h[n]) == d and int(info_h[n]) == inf["info"] and rew_h[n] == np.float32(r), (t, i)
            for k in sub:
                assert np.array_equal(sub[k][n].reshape(ob[k].shape), ob[k].astype(np.float32)), (k, t, i)
            nd += d
    env.close()
    print("soak %s: E=%d, %d steps, %d sampled envs bit-exact vs oracle (%d episode ends among them) in %.1f s" % (kw, E, T, len(sample), nd, time.time() - t0))
soak(dict(human_num=20), 4096, 1500, 40)
soak(dict(human_num=50, randomize_attributes=1, random_goal_changing=1), 2048, 400, 24)
soak(dict(human_num=20, env_kind=1, phase=2), 1024, 600, 24)
soak(dict(human_num=10, robot_policy=1, randomize_attributes=1, random_goal_changing=1, phase=2), 1024, 600, 24)
soak(dict(human_num=15, human_num_range=5, randomize_attributes=1, random_goal_changing=1), 2048, 600, 24)
soak(dict(human_num=6, human_num_range=5, kinematics=1, randomize_attributes=1), 2048, 600, 24)
soak(dict(human_num=20, humans_policy=1, robot_visible=1), 2048, 500, 24)
soak(dict(human_num=20, env_kind=1, predict_truth=1), 1024, 400, 16)


def soak_update(E, T, updates):
    """The training loop (rollout + GAE + PPO.update) `updates` times, twice from the same seed: losses and every weight bit-identical --
    the update path's counterpart of the simulator soak above (a size-dependent scratch overrun in cn_rn_seq_bwd showed up 1-in-8 in round 4)."""
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.trainer import train
    t0 = time.time()
    runs = []
    for _ in range(2):
        hist, pol = train("CrowdSimVarNum-v0", num_processes=E, num_steps=T, num_updates=updates, seed=31, config=C.non_randomized(), log=None)
        runs.append(({k: v.detach().clone() for k, v in pol.state_dict().items()}, [(r["value_loss"], r["action_loss"]) for r in hist]))
    assert runs[0][1] == runs[1][1], "losses differ between two runs from the same seed"
    for k, v in runs[0][0].items():
        assert torch.equal(v, runs[1][0][k]), k
    print("soak update: E=%d T=%d, %d updates twice, bit-identical (%.1f s)" % (E, T, updates, time.time() - t0))


soak_update(8, 5, 50)
soak_update(64, 8, 50)
soak_update(4096, 30, 50)
