"""-m gpu: deterministic evaluation (rl/evaluation.py protocol) in the test phase -- the reference-shaped sequential
loop over ONE env and the batched form (every test case one env) must report the same metrics."""
import logging

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("env_name,over", [("CrowdSimVarNum-v0", {}), ("CrowdSimPred-v0", {"sim.predict_method": "const_vel"})])
def test_sequential_and_batched_evaluation_agree(env_name, over):
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.evaluation import evaluate, evaluate_batched
    from crowdnav_prediction_attngraph_amd.policy import Policy
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    cfg = C.non_randomized(**dict({"sim.human_num": 10, "env.test_size": 16}, **over))
    dev = torch.device("cuda", 0)
    envs = make_vec_envs(env_name, 7, 1, 0.99, None, dev, True, config=cfg)          # 1 env -> phase 'test' (envs.py:55-58)
    assert envs.cfg.phase == 2
    torch.manual_seed(3)
    pol = Policy(envs.observation_space.spaces, envs.action_space, base="selfAttn_merge_srnn",
                 base_kwargs=dict(env_name=env_name, num_processes=1, num_mini_batch=1, seq_length=30)).to(dev)
    with torch.no_grad():                      # make the untrained policy head for the goal-ish direction: varied outcomes
        pol.dist.fc_mean.bias.copy_(torch.tensor([0.3, -0.2]))
    log = logging.getLogger("eval-test")
    n = 20                                     # > test_size / 2: the case index wraps and cases repeat, as in the reference
    # bit-identical episodes need launches whose per-env result does not depend on the batch: batch_invariant=True (the default runs the
    # fused kernels training and bench.py use, checked below through the fraction of episodes that end the same way)
    seq = evaluate(pol, envs, 1, dev, n, log, cfg, None, batch_invariant=True)
    bat = evaluate_batched(pol, env_name, cfg, 7, n, device=dev, logging=log, batch_invariant=True)
    assert seq["episodes"] == bat["episodes"] == n
    for k in ("success_rate", "collision_rate", "timeout_rate", "collision_cases", "timeout_cases"):
        assert seq[k] == bat[k], (k, seq[k], bat[k])
    for k in ("nav_time", "path_length", "intrusion_ratio", "mean_reward"):
        assert seq[k] == pytest.approx(bat[k], rel=1e-6, abs=1e-6), (k, seq[k], bat[k])
    if seq["min_intrusion_dist"] == seq["min_intrusion_dist"]:
        assert seq["min_intrusion_dist"] == pytest.approx(bat["min_intrusion_dist"], rel=1e-9)
    assert seq["collision_rate"] + seq["timeout_rate"] + seq["success_rate"] == pytest.approx(1.0)
    # default mode = the fused rollout kernels: the same protocol, episodes may differ from the invariant run only through the 1e-7-level
    # neighbour sensitivity of the policy output (a chaotic episode can flip): the three rates stay within 2 episodes of 20
    fused = evaluate_batched(pol, env_name, cfg, 7, n, device=dev, logging=log)
    assert pol.rollout_gemm_mode == "fused" and fused["episodes"] == n
    for k in ("success_rate", "collision_rate", "timeout_rate"):
        assert abs(fused[k] - bat[k]) <= 2.0 / n + 1e-9, (k, fused[k], bat[k])


@pytest.mark.parametrize("fixture,robot", [("ref_eval_orca_robot_log.json", "orca"), ("ref_eval_sf_robot_log.json", "social_force")])
def test_device_evaluation_reproduces_the_shipped_scripted_robot_logs(fixture, robot):
    """The reference's own end-to-end fixtures (trained_models/ORCA_no_rand and SF_no_rand test logs, real Python-RVO2), replayed
    through the reference-shaped `evaluate` on the HIP simulator: all 500 outcomes and the six logged metrics."""
    import json
    import os
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.evaluation import evaluate
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture)))
    c = ref["config"]
    cfg = C.Config(**{"sim.human_num": c["human_num"], "robot.policy": robot, "env.randomize_attributes": bool(c["randomize_attributes"]),
                      "humans.random_goal_changing": bool(c["random_goal_changing"]), "humans.end_goal_changing": bool(c["end_goal_changing"]),
                      "env.test_size": c["test_size"]})
    dev = torch.device("cuda", 0)
    envs = make_vec_envs(c["env_name"], c["seed"], 1, 0.99, None, dev, True, config=cfg)
    m = evaluate(None, envs, 1, dev, c["test_size"], logging.getLogger("eval-test"), cfg, None)
    assert m["collision_cases"] == ref["collision_cases"]
    assert m["timeout_cases"] == ref["timeout_cases"]
    for k in ("success_rate", "collision_rate", "timeout_rate", "nav_time", "path_length", "intrusion_ratio", "min_intrusion_dist"):
        assert "%.2f" % m[k] == "%.2f" % ref[k], (k, m[k], ref[k])


def test_batched_evaluation_with_orca_robot_matches_sequential():
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.evaluation import evaluate, evaluate_batched
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    cfg = C.non_randomized(**{"sim.human_num": 20, "robot.policy": "orca", "env.test_size": 40})
    dev = torch.device("cuda", 0)
    log = logging.getLogger("eval-test")
    seq = evaluate(None, make_vec_envs("CrowdSimVarNum-v0", 425, 1, 0.99, None, dev, True, config=cfg), 1, dev, 30, log, cfg, None)
    bat = evaluate_batched(None, "CrowdSimVarNum-v0", cfg, 425, 30, device=dev, logging=log)
    for k in ("success_rate", "collision_rate", "timeout_rate", "collision_cases", "timeout_cases"):
        assert seq[k] == bat[k], (k, seq[k], bat[k])
    for k in ("nav_time", "path_length", "intrusion_ratio", "min_intrusion_dist"):
        assert seq[k] == pytest.approx(bat[k], rel=1e-6), (k, seq[k], bat[k])
    assert seq["success_rate"] > 0.3          # the ORCA robot does reach goals
