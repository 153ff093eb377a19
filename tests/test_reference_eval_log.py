"""The reference ships two end-to-end results produced with the real Python-RVO2: the evaluation logs of its ORCA-driven robot
(trained_models/ORCA_no_rand/test/test_00000.pt.log, randomised humans) and of its social-force robot
(trained_models/SF_no_rand/test/test_00000.pt.log, the non-randomised humans of BASELINE configs[1]); each holds 500 test
episodes with per-episode outcome lists and six aggregate metrics.
Replaying test.py's protocol with the CPU oracle -- one env for the whole run (the robot's rvo2 simulator, with the radii and
neighbour distance frozen at its creation, lives across episodes), `reset()` per episode on top of the vec-env auto-reset,
robot action = ORCA on the robot's beliefs, Danger from the humans' true future positions -- must reproduce that log exactly.

Crowd dynamics are chaotic: one differing rounding in the ORCA linear programs flips collision / success outcomes within a few
steps, so agreement on all 500 outcomes (146 collisions, 8 timeouts at exactly the logged episode indices) pins the RVO2
restatement, the scenario RNG stream, the randomisation quirks and the test-phase logic against the reference's own fixture.
"""
import json
import os

import numpy as np

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def replay(cfg, seed, n_episodes):
    env = O.OracleEnv(cfg, seed)
    out = []
    for _ in range(n_episodes):
        ob = env.reset()
        last = ob["robot_node"][0, :2].copy()
        path, n, close, mins = 0.0, 0, 0, []
        while True:
            n += 1
            ob, _, done, info = env.step([0.0, 0.0], autoreset=True)     # the action is ignored: robot.policy == 'orca'
            pos = ob["robot_node"][0, :2]
            path += float(np.linalg.norm(pos - last))                     # rl/evaluation.py:93-94 (incl. the jump to the auto-reset start)
            last = pos.copy()
            if info["info"] == 4:
                close += 1
                mins.append(info["min_dist"])
            if done:
                break
        out.append((info["info"], n, path, close, mins))
    return out


import pytest  # noqa: E402


@pytest.mark.parametrize("fixture,robot_policy", [("ref_eval_orca_robot_log.json", 1), ("ref_eval_sf_robot_log.json", 2)])
def test_oracle_reproduces_the_shipped_evaluation_logs(fixture, robot_policy):
    ref = json.load(open(os.path.join(HERE, "golden", fixture)))
    c = ref["config"]
    cfg = O.default_config(human_num=c["human_num"], phase=2, robot_policy=robot_policy, nenv=1, randomize_attributes=c["randomize_attributes"],
                           random_goal_changing=c["random_goal_changing"], end_goal_changing=c["end_goal_changing"], test_size=c["test_size"])
    N = c["test_size"]
    out = replay(cfg, c["seed"], N)
    coll = [k for k in range(N) if out[k][0] == 2]
    tout = [k for k in range(N) if out[k][0] == 1]
    succ = [k for k in range(N) if out[k][0] == 3]
    assert coll == ref["collision_cases"]
    assert tout == ref["timeout_cases"]
    assert len(succ) + len(coll) + len(tout) == N
    assert len(coll) > 100 and len(tout) >= 8
    # the six aggregates, formatted like rl/evaluation.py:141-146
    got = dict(success_rate=len(succ) / N, collision_rate=len(coll) / N, timeout_rate=len(tout) / N,
               nav_time=np.mean([(out[k][1] - 1) * 0.25 for k in succ]), path_length=np.mean([o[2] for o in out]),
               intrusion_ratio=np.mean([100.0 * o[3] / o[1] for o in out]), min_intrusion_dist=np.mean([m for o in out for m in o[4]]))
    for k, v in got.items():
        assert "%.2f" % v == "%.2f" % ref[k], (k, v, ref[k])
    # episodes k and k + 250 are the same test case (two resets per episode, case index modulo test_size)
    assert all(out[k][:2] == out[k + 250][:2] for k in range(250))
