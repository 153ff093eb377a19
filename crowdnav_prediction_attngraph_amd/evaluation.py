"""Deterministic evaluation (drop-in for rl/evaluation.py:7-160) on the batched device simulator.

`evaluate` keeps the reference's argument list and its sequential semantics (one env, `test_size` episodes, a
`reset()` per episode on top of the vec-env's own auto-reset -- so consecutive episodes use every SECOND test case and the
case index wraps at `test_size`; the robot-path length includes the jump to the auto-reset start position; nav time is the
env time before the final step).  `evaluate_batched` produces the same numbers by running every distinct test case as one
env of a single batch (first episode of each env), which is how the device simulator wants to be driven.
Both return the metrics as a dict (the reference only logs them).
"""
import numpy as np
import torch

from . import info as I
from .config import to_env_config


def _summarise(outcomes, steps, path_len, too_close, min_dists, ep_rewards, time_limit, time_step, logging):
    test_size = len(outcomes)
    success = [k for k, o in enumerate(outcomes) if o == 3]
    collision = [k for k, o in enumerate(outcomes) if o == 2]
    timeout = [k for k, o in enumerate(outcomes) if o == 1]
    assert len(success) + len(collision) + len(timeout) == test_size
    success_times = [(steps[k] - 1) * time_step for k in success]      # env.global_time read before the final step (:75-76)
    m = dict(success_rate=len(success) / test_size, collision_rate=len(collision) / test_size, timeout_rate=len(timeout) / test_size,
             nav_time=sum(success_times) / len(success_times) if success_times else time_limit,
             path_length=float(np.mean(path_len)), intrusion_ratio=float(np.mean([100.0 * c / s for c, s in zip(too_close, steps)])),
             min_intrusion_dist=float(np.mean(min_dists)) if len(min_dists) else float("nan"),
             collision_cases=collision, timeout_cases=timeout, mean_reward=float(np.mean(ep_rewards)), episodes=test_size)
    if logging is not None:
        logging.info('Testing success rate: {:.2f}, collision rate: {:.2f}, timeout rate: {:.2f}, '
                     'nav time: {:.2f}, path length: {:.2f}, average intrusion ratio: {:.2f}%, '
                     'average minimal distance during intrusions: {:.2f}'.format(m["success_rate"], m["collision_rate"], m["timeout_rate"],
                                                                                  m["nav_time"], m["path_length"], m["intrusion_ratio"],
                                                                                  m["min_intrusion_dist"]))
        logging.info('Collision cases: ' + ' '.join(str(x) for x in collision))
        logging.info('Timeout cases: ' + ' '.join(str(x) for x in timeout))
    return m


class _batch_invariant(object):
    """batch_invariant=True: run the policy with the launches whose per-env results do not depend on which other envs share the batch
    ('bf16x3': the same arithmetic as separate launches), so that one-env-at-a-time and all-cases-in-one-batch evaluation give
    bit-identical episodes.  Default (False): the policy's own rollout mode -- 'fused' unless the caller changed it, i.e. exactly the
    arithmetic training and bench.py run; its per-env results depend on the tile neighbours at the 1e-7 level (softmax summation order)."""

    def __init__(self, actor_critic, on):
        self.ac = actor_critic if on else None

    def __enter__(self):
        if self.ac is not None and hasattr(self.ac, "rollout_gemm_mode"):
            self.saved = self.ac.rollout_gemm_mode
            self.ac.rollout_gemm_mode = "bf16x3"
        return self

    def __exit__(self, *exc):
        if self.ac is not None and hasattr(self.ac, "rollout_gemm_mode"):
            self.ac.rollout_gemm_mode = self.saved
        return False


def evaluate(actor_critic, eval_envs, num_processes, device, test_size, logging, config, args, visualize=False, *, batch_invariant=False):
    """Same call as the reference's `evaluate`.  eval_envs = make_vec_envs(..., num_processes=1, ...) (phase 'test').  The policy runs in its
    own rollout mode (the benchmarked fused kernels) unless batch_invariant=True is passed (see _batch_invariant)."""
    with _batch_invariant(actor_critic, batch_invariant):
        return _evaluate(actor_critic, eval_envs, num_processes, device, test_size, logging, config, args, visualize)


def _evaluate(actor_critic, eval_envs, num_processes, device, test_size, logging, config, args, visualize=False):
    if num_processes != 1 or eval_envs.num_envs != 1:
        raise NotImplementedError("the reference evaluates with ONE env (test.py:136); use evaluate_batched for the parallel form")
    if visualize:
        # rl/evaluation.py:84-85 draws every step.  Rendering is out of scope here, but the reference's test.py cannot switch it off (its
        # --visualize flag is `default=True, action='store_true'`, test.py:25), so the request is acknowledged instead of refused: the
        # episodes run, nothing is drawn, the metrics are the same.
        import warnings
        warnings.warn("evaluate(visualize=True): rendering is not implemented on the accelerated path; running the episodes without drawing")
    scripted = actor_critic is None          # robot.policy in ('orca', ...): the env drives the robot itself (test.py:152-153)
    if scripted and int(eval_envs.cfg.robot_policy) == 0:
        raise ValueError("actor_critic is None but the env was not configured with robot.policy = 'orca'")
    if not scripted:
        base = actor_critic.base
        hxs = {"human_node_rnn": torch.zeros(1, 1, base.human_node_rnn_size, device=device),
               "human_human_edge_rnn": torch.zeros(1, base.human_num + 1, base.human_human_edge_rnn_size, device=device)}
    masks = torch.zeros(1, 1, device=device)
    time_limit, time_step = float(eval_envs.cfg.time_limit), float(eval_envs.cfg.time_step)
    outcomes, steps, path_lens, too_closes, min_dists, ep_rewards = [], [], [], [], [], []
    for _ in range(test_size):
        obs = eval_envs.reset()
        done, n, too_close, path_len = False, 0, 0, 0.0
        last_pos = obs["robot_node"][0, 0, :2].cpu().numpy()
        while not done:
            n += 1
            if scripted:
                action = torch.zeros(1, 2, device=device)                       # rl/evaluation.py:73-74
            else:
                with torch.no_grad():
                    _, action, _, hxs = actor_critic.act(obs, hxs, masks, deterministic=True)
            obs, rew, dones, infos = eval_envs.step(action)
            pos = obs["robot_node"][0, 0, :2].cpu().numpy()
            path_len += float(np.linalg.norm(pos - last_pos))
            last_pos = pos
            if isinstance(infos[0]["info"], I.Danger):
                too_close += 1
                min_dists.append(infos[0]["info"].min_dist)
            done = bool(dones[0])
            masks = torch.tensor([[0.0] if done else [1.0]], dtype=torch.float32, device=device)
            if "episode" in infos[0]:
                ep_rewards.append(infos[0]["episode"]["r"])
        code = {I.Timeout: 1, I.Collision: 2, I.ReachGoal: 3}.get(type(infos[0]["info"]))
        if code is None:
            raise ValueError("Invalid end signal from environment")
        outcomes.append(code); steps.append(n); path_lens.append(path_len); too_closes.append(too_close)
    eval_envs.close()
    return _summarise(outcomes, steps, path_lens, too_closes, min_dists, ep_rewards, time_limit, time_step, logging)


def evaluate_batched(actor_critic, env_name, config, seed, test_size, device=None, logging=None, *, batch_invariant=False):
    """The same protocol with every distinct test case as one env of one batch (all tensors stay on the GPU).  batch_invariant=True on both
    this and evaluate() makes the two report bit-identical episodes (tests/test_gpu_eval.py); by default both run the fused kernels and
    agree to the policy's 1e-7-level sensitivity to its tile neighbours (which can flip a chaotic episode's outcome)."""
    if env_name == "CrowdSimPredRealGST-v0":
        # the raw env observation carries placeholder futures; the policy needs the VecPretextNormalize processing (GST predictions,
        # distance sort, social penalty), which this function does not run
        raise NotImplementedError("evaluate_batched does not run the GST wrapper: evaluate CrowdSimPredRealGST-v0 with "
                                  "evaluate(actor_critic, make_vec_envs(..., pretext_wrapper=True), ...)")
    with _batch_invariant(actor_critic, batch_invariant):
        return _evaluate_batched(actor_critic, env_name, config, seed, test_size, device, logging)


def _evaluate_batched(actor_critic, env_name, config, seed, test_size, device=None, logging=None):
    from .hip import HipEnvBatch
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    cfg = to_env_config(config, env_name, 1, "test")          # nenv = 1: case counters advance by one, as in the sequential run
    scripted = actor_critic is None
    if scripted and int(cfg.robot_policy) == 0:
        raise ValueError("actor_critic is None but config.robot.policy is not 'orca'")
    if int(cfg.robot_policy) == 1 and int(cfg.randomize_attributes):
        raise NotImplementedError("an ORCA robot with randomised humans keeps ONE rvo2 simulator (radii / neighbour distance frozen in the "
                                  "first episode) across the whole sequential run; use evaluate() for that configuration")
    case_size = int(cfg.test_size)
    case_of = [(2 * k) % case_size for k in range(test_size)]  # reset() + the vec-env's auto-reset: two resets per episode
    cases = sorted(set(case_of))
    E = len(cases)
    env = HipEnvBatch(cfg, E, int(seed), device=device)
    # env e draws seed offset + counter[e] + (seed + e): counter[e] = case - e  (cases are distinct and sorted, so case >= e)
    env.set_case_counters(torch.tensor([c - e for e, c in enumerate(cases)], dtype=torch.int64))
    obs = env.reset()
    pol = None if scripted else actor_critic._hip_policy(E, device)
    zero_action = torch.zeros(E, 2, device=device)
    H = env.H
    hx = [torch.zeros(E, 1, 128, device=device), torch.zeros(E, 1, 128, device=device)]
    masks = torch.zeros(E, 1, device=device)
    active = torch.ones(E, dtype=torch.bool, device=device)
    steps = torch.zeros(E, dtype=torch.int64, device=device)
    too_close = torch.zeros(E, dtype=torch.int64, device=device)
    path_len = torch.zeros(E, dtype=torch.float64, device=device)
    outcome = torch.zeros(E, dtype=torch.int64, device=device)
    ep_reward = torch.zeros(E, dtype=torch.float64, device=device)
    last_pos = obs["robot_node"].view(E, 7)[:, :2].clone()
    md_steps = []                                              # (env, step, min_dist) of every Danger step of a first episode
    max_steps = int(round(float(cfg.time_limit) / float(cfg.time_step))) + 1
    for t in range(max_steps):
        if scripted:
            action = zero_action
        else:
            pobs = {k: obs[k] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
            out = pol.act(pobs, hx[t & 1], masks, eps=None)      # deterministic: dist.mode() (model.py:66-67)
            hx[(t + 1) & 1] = out["hxs"].view(E, 1, 128)
            action = out["action"]
        obs, rew, done, info, ep_ret, _ = env.step(action)
        pos = obs["robot_node"].view(E, 7)[:, :2]
        # the reference measures the path on the float32 observation tensors, episode-end jump to the auto-reset start included
        path_len += torch.where(active, torch.linalg.norm((pos - last_pos).float(), dim=1).double(), torch.zeros_like(path_len))
        last_pos = pos.clone()
        steps += active.long()
        danger = active & (info == 4)
        if bool(danger.any()):
            md = env.get_danger_min_dist()
            idx = danger.nonzero().squeeze(1)
            md_steps.append(torch.stack([idx.double(), torch.full_like(idx, t).double(), md[idx]], 1))
        too_close += danger.long()
        fin = active & (done != 0)
        outcome = torch.where(fin, info.long(), outcome)
        ep_reward = torch.where(fin, ep_ret, ep_reward)
        active = active & ~fin
        masks = (done == 0).float().view(E, 1)
        if not bool(active.any()):
            break
    env.close()
    outcome_h, steps_h, path_h, close_h = outcome.cpu().tolist(), steps.cpu().tolist(), path_len.cpu().tolist(), too_close.cpu().tolist()
    rew_h = [round(x, 6) for x in ep_reward.cpu().tolist()]
    md = torch.cat(md_steps).cpu().numpy() if md_steps else np.zeros((0, 3))
    e_of = {c: e for e, c in enumerate(cases)}
    outcomes, steps_l, paths, closes, mins, rewards = [], [], [], [], [], []
    for k in range(test_size):                                  # sequential episode k == first episode of the env of its case
        e = e_of[case_of[k]]
        outcomes.append(outcome_h[e]); steps_l.append(steps_h[e]); paths.append(path_h[e]); closes.append(close_h[e]); rewards.append(rew_h[e])
        rows = md[md[:, 0] == e]
        mins.extend(rows[np.argsort(rows[:, 1]), 2].tolist())
    return _summarise(outcomes, steps_l, paths, closes, mins, rewards, float(cfg.time_limit), float(cfg.time_step), logging)
