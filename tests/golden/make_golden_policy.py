#!/usr/bin/env python3
"""Golden vectors for the policy forward (act / evaluate_actions), GAE, advantage normalisation and one PPO.update,
produced by the reference's own torch code (build container only; see _ref_import.py)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_import as R  # noqa: E402
import policy_util as PU  # noqa: E402


def ref_args(env_name, E, nmb, T):
    from arguments import get_args
    a = get_args()
    a.env_name = env_name
    a.num_processes = E
    a.num_mini_batch = nmb
    a.seq_length = T
    a.num_steps = T
    a.no_cuda = True
    a.cuda = False
    return a


def spaces(H, D):
    import gym
    d = {"robot_node": gym.spaces.Box(-np.inf, np.inf, (1, 7)), "temporal_edges": gym.spaces.Box(-np.inf, np.inf, (1, 2)),
         "spatial_edges": gym.spaces.Box(-np.inf, np.inf, (H, D)), "detected_human_num": gym.spaces.Box(-np.inf, np.inf, (1,)),
         "visible_masks": gym.spaces.Box(-np.inf, np.inf, (H,), dtype=bool)}
    act = gym.spaces.Box(-np.inf * np.ones(2), np.inf * np.ones(2))
    act.__class__.__name__ = "Box"
    return gym.spaces.Dict(d), act


def build_policy(env_name, E, H, D, nmb=1, T=1):
    import torch
    from rl.networks.model import Policy
    args = ref_args(env_name, E, nmb, T)
    ob_space, act_space = spaces(H, D)
    torch.manual_seed(0)
    pol = Policy(ob_space.spaces, act_space, base_kwargs=args, base="selfAttn_merge_srnn")
    sd = pol.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    new = PU.formula_state_dict(shapes)
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
    return pol, args, shapes, ob_space, act_space


def t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x))


def policy_case(tag, env_name, E, H, D):
    import torch
    pol, args, shapes, _, _ = build_policy(env_name, E, H, D)
    ob = PU.synth_obs(E, H, D, seed=1000 + E * 7 + H)
    rs = np.random.RandomState(5)
    hxs = {"human_node_rnn": rs.uniform(-1, 1, (E, 1, 128)).astype(np.float32),
           "human_human_edge_rnn": np.zeros((E, H + 1, 256), np.float32)}
    masks = np.ones((E, 1), np.float32)
    if E > 1:
        masks[1] = 0.0
    taps = {}
    base = pol.base
    hooks = [base.spatial_attn.register_forward_hook(lambda m, i, o: taps.__setitem__("hh_out", o.detach().numpy().copy())),
             base.spatial_linear.register_forward_hook(lambda m, i, o: taps.__setitem__("spatial_lin", o.detach().numpy().copy())),
             base.attn.register_forward_hook(lambda m, i, o: taps.update(hr_out=o[0].detach().numpy().copy(),
                                                                         hr_attn=o[1].detach().numpy().copy())),
             base.robot_linear.register_forward_hook(lambda m, i, o: taps.__setitem__("robot_emb", o.detach().numpy().copy()))]
    with torch.no_grad():
        tob = {k: t(v) for k, v in ob.items()}
        value, action, logp, hx_out = pol.act(tob, {k: t(v) for k, v in hxs.items()}, t(masks), deterministic=True)
        _, actor_feat, _ = pol.base(tob, {k: t(v) for k, v in hxs.items()}, t(masks), infer=True)
        # stochastic log-prob check at a fixed action
        fixed_action = rs.uniform(-1.5, 1.5, (E, 2)).astype(np.float32)
        dist = pol.dist(actor_feat)
        logp_fixed = dist.log_probs(t(fixed_action))
        entropy = dist.entropy().mean()
    for h in hooks:
        h.remove()
    out = dict(ob)
    out.update(hxs_node=hxs["human_node_rnn"], masks=masks, value=value.numpy(), action=action.numpy(), logp=logp.numpy(),
               hx_out=hx_out["human_node_rnn"].numpy(), actor_feat=actor_feat.numpy(), fixed_action=fixed_action,
               logp_fixed=logp_fixed.numpy(), entropy=np.float32(entropy.item()),
               hh_out=taps["hh_out"].reshape(E, H, 512), spatial_lin=taps["spatial_lin"].reshape(E, H, 256),
               hr_out=taps["hr_out"].reshape(E, 256), hr_attn=taps["hr_attn"].reshape(E, H), robot_emb=taps["robot_emb"].reshape(E, 256),
               meta=np.array(json.dumps(dict(env_name=env_name, E=E, H=H, D=D, shapes={k: list(v) for k, v in shapes.items()}))))
    path = os.path.join(HERE, "policy_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print("policy %-16s value[0]=%.5f action[0]=%s -> %s (%.0f KB)" % (tag, out["value"][0, 0], out["action"][0], os.path.basename(path),
                                                                   os.path.getsize(path) / 1024))


def variant_obs(E, H, D, seed, unsorted):
    """synth_obs, and for sort_humans = False: humans in random order with the visibility as a mask (one sample with nobody visible)."""
    ob = PU.synth_obs(E, H, D, seed=seed)
    rs = np.random.RandomState(seed + 17)
    det = ob["detected_human_num"].reshape(E).astype(int)
    vm = np.arange(H)[None, :] < det[:, None]
    if unsorted:
        for e in range(E):
            perm = rs.permutation(H)
            ob["spatial_edges"][e] = ob["spatial_edges"][e][perm]
            vm[e] = vm[e][perm]
        vm[E // 2] = False                       # nobody visible: the reference then keeps human 0 (selfAttn_srnn_temp_node.py:381-383)
        ob["detected_human_num"] = np.maximum(vm.sum(1), 1).astype(np.float32).reshape(E, 1)
    ob["visible_masks"] = vm
    return ob


def variant_case(tag, env_name, N, H, D, T, use_self_attn, sort_humans):
    """args.use_self_attn = False and / or args.sort_humans = False (arguments.py:189, :206): act on N envs and evaluate_actions on a [T, N]
    slice, by the reference's own modules."""
    import torch
    from rl.networks.model import Policy
    args = ref_args(env_name, N, 1, T)
    args.use_self_attn, args.sort_humans = use_self_attn, sort_humans
    ob_space, act_space = spaces(H, D)
    torch.manual_seed(0)
    pol = Policy(ob_space.spaces, act_space, base_kwargs=args, base="selfAttn_merge_srnn")
    init_sum = {k: float(v.double().abs().sum()) for k, v in pol.state_dict().items()}      # seeded-init parity of the variant's modules
    shapes = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in PU.formula_state_dict(shapes).items()})
    rs = np.random.RandomState(9)
    obs_seq = variant_obs(T * N, H, D, 4000 + H + 10 * D, not sort_humans)
    hxs = {"human_node_rnn": rs.uniform(-1, 1, (N, 1, 128)).astype(np.float32), "human_human_edge_rnn": np.zeros((N, H + 1, 256), np.float32)}
    masks_seq = (rs.uniform(size=(T * N, 1)) > 0.2).astype(np.float32)
    actions = rs.uniform(-1.5, 1.5, (T * N, 2)).astype(np.float32)
    taps = {}
    hooks = [pol.base.spatial_linear.register_forward_hook(lambda m, i, o: taps.__setitem__("spatial_lin", o.detach().numpy().copy())),
             pol.base.attn.register_forward_hook(lambda m, i, o: taps.update(hr_out=o[0].detach().numpy().copy(), hr_attn=o[1].detach().numpy().copy()))]
    with torch.no_grad():
        first = {k: t(v[:N]) for k, v in obs_seq.items()}
        value, action, logp, hx_out = pol.act(first, {k: t(v) for k, v in hxs.items()}, t(masks_seq[:N]), deterministic=True)
        act_taps = dict(taps)
        ev_value, ev_logp, ev_ent, ev_hx = pol.evaluate_actions({k: t(v) for k, v in obs_seq.items()}, {k: t(v) for k, v in hxs.items()}, t(masks_seq), t(actions))
    for h in hooks:
        h.remove()
    out = {"obs_" + k: v for k, v in obs_seq.items()}
    out.update(hxs_node=hxs["human_node_rnn"], masks=masks_seq, actions=actions, value=value.numpy(), action=action.numpy(), logp=logp.numpy(),
               hx_out=hx_out["human_node_rnn"].numpy(), spatial_lin=act_taps["spatial_lin"].reshape(N, H, 256), hr_out=act_taps["hr_out"].reshape(N, 256),
               hr_attn=act_taps["hr_attn"].reshape(N, H), ev_value=ev_value.numpy(), ev_logp=ev_logp.numpy(), ev_entropy=np.float32(ev_ent.item()),
               ev_hx=ev_hx["human_node_rnn"].numpy(),
               meta=np.array(json.dumps(dict(env_name=env_name, N=N, H=H, D=D, T=T, use_self_attn=use_self_attn, sort_humans=sort_humans,
                                             shapes={k: list(v) for k, v in shapes.items()}, init_abs_sum=init_sum))))
    path = os.path.join(HERE, "polvar_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print("variant %-22s value[0]=%.5f -> %s (%.0f KB)" % (tag, out["value"][0, 0], os.path.basename(path), os.path.getsize(path) / 1024))


def rollout_variant_case(tag, env_name, E, H, D, T, nmb, use_self_attn, sort_humans):
    """One PPO.update of the reference with args.use_self_attn / args.sort_humans switched (arguments.py:189, :206): a rollout stored with its
    visibility masks, three losses and every post-update tensor sampled at <= 256 positions."""
    import torch
    from rl.networks.model import Policy
    from rl.networks.storage import RolloutStorage
    from rl import ppo as ref_ppo
    args = ref_args(env_name, E, nmb, T)
    args.use_self_attn, args.sort_humans = use_self_attn, sort_humans
    ob_space, act_space = spaces(H, D)
    torch.manual_seed(0)
    pol = Policy(ob_space.spaces, act_space, base_kwargs=args, base="selfAttn_merge_srnn")
    shapes = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in PU.formula_state_dict(shapes).items()})
    rollouts = RolloutStorage(T, E, ob_space.spaces, act_space, 128, 256)
    rs = np.random.RandomState(23)
    obs_seq = [variant_obs(E, H, D, 700 + s, not sort_humans) for s in range(T + 1)]
    dones = rs.uniform(size=(T, E)) < 0.15
    rewards = rs.uniform(-1, 1, (T, E, 1)).astype(np.float32)
    for k in rollouts.obs:
        rollouts.obs[k][0].copy_(t(obs_seq[0][k]))
    torch.manual_seed(123)
    for s in range(T):
        with torch.no_grad():
            ob = {k: rollouts.obs[k][s] for k in rollouts.obs}
            hx = {k: rollouts.recurrent_hidden_states[k][s] for k in rollouts.recurrent_hidden_states}
            value, action, logp, hx_new = pol.act(ob, hx, rollouts.masks[s])
        masks = t(np.where(dones[s], 0.0, 1.0).astype(np.float32).reshape(E, 1))
        rollouts.insert({k: t(v) for k, v in obs_seq[s + 1].items()}, hx_new, action, logp, value, t(rewards[s]), masks, torch.ones(E, 1))
    with torch.no_grad():
        ob = {k: rollouts.obs[k][-1] for k in rollouts.obs}
        hx = {k: rollouts.recurrent_hidden_states[k][-1] for k in rollouts.recurrent_hidden_states}
        next_value = pol.get_value(ob, hx, rollouts.masks[-1]).detach()
    rollouts.compute_returns(next_value, True, 0.99, 0.95, False)
    out = dict(rewards=rewards, actions=rollouts.actions.numpy().copy(), logp=rollouts.action_log_probs.numpy().copy(), values=rollouts.value_preds.numpy().copy(),
               next_value=next_value.numpy(), returns=rollouts.returns.numpy().copy(), masks=rollouts.masks.numpy().copy(),
               hxs_node=rollouts.recurrent_hidden_states["human_node_rnn"].numpy().copy())
    agent = ref_ppo.PPO(pol, 0.2, 2, nmb, 0.5, 0.0, lr=4e-5, eps=1e-5, max_grad_norm=0.5)
    torch.manual_seed(321)
    v_loss, a_loss, ent = agent.update(rollouts)
    out["losses"] = np.array([v_loss, a_loss, ent], dtype=np.float64)
    for k, v in pol.state_dict().items():
        flat = v.detach().numpy().reshape(-1)
        out["smp_" + k] = flat[np.linspace(0, flat.size - 1, min(flat.size, 256)).astype(np.int64)].copy()
        out["chk_" + k] = np.array([float(flat.astype(np.float64).sum()), float(np.abs(flat.astype(np.float64)).sum())])
    for s in range(T + 1):
        for k, v in obs_seq[s].items():
            out["obs%d_%s" % (s, k)] = v
    out["meta"] = np.array(json.dumps(dict(env_name=env_name, E=E, N=E, H=H, D=D, T=T, nmb=nmb, use_self_attn=use_self_attn, sort_humans=sort_humans, update_seed=321,
                                           ppo_epoch=2, shapes={k: list(v) for k, v in shapes.items()})))
    path = os.path.join(HERE, "rollvar_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print("rollout variant %-22s losses=%s -> %s (%.0f KB)" % (tag, out["losses"], os.path.basename(path), os.path.getsize(path) / 1024))


def rollout_case(tag, env_name, E, H, D, T, nmb):
    """A synthetic rollout pushed through the reference RolloutStorage / compute_returns / PPO.update."""
    import torch
    from rl.networks.storage import RolloutStorage
    from rl import ppo as ref_ppo
    pol, args, shapes, ob_space, act_space = build_policy(env_name, E, H, D, nmb=nmb, T=T)
    rollouts = RolloutStorage(T, E, ob_space.spaces, act_space, 128, 256)
    rs = np.random.RandomState(11)
    obs_seq = [PU.synth_obs(E, H, D, seed=200 + s) for s in range(T + 1)]
    dones = rs.uniform(size=(T, E)) < 0.15
    rewards = rs.uniform(-1, 1, (T, E, 1)).astype(np.float32)
    for k in rollouts.obs:
        if k in obs_seq[0]:
            rollouts.obs[k][0].copy_(t(obs_seq[0][k]))
    # act with the reference policy so log-probs / values / hidden states are self-consistent
    torch.manual_seed(123)
    actions_rec, logp_rec, value_rec = [], [], []
    for s in range(T):
        with torch.no_grad():
            ob = {k: rollouts.obs[k][s] for k in rollouts.obs}
            hx = {k: rollouts.recurrent_hidden_states[k][s] for k in rollouts.recurrent_hidden_states}
            value, action, logp, hx_new = pol.act(ob, hx, rollouts.masks[s])
        masks = t(np.where(dones[s], 0.0, 1.0).astype(np.float32).reshape(E, 1))
        nxt = {k: t(obs_seq[s + 1][k]) for k in obs_seq[s + 1]}
        nxt["visible_masks"] = torch.zeros(E, H, dtype=torch.bool)
        rollouts.insert(nxt, hx_new, action, logp, value, t(rewards[s]), masks, torch.ones(E, 1))
        actions_rec.append(action.numpy().copy()); logp_rec.append(logp.numpy().copy()); value_rec.append(value.numpy().copy())
    with torch.no_grad():
        ob = {k: rollouts.obs[k][-1] for k in rollouts.obs}
        hx = {k: rollouts.recurrent_hidden_states[k][-1] for k in rollouts.recurrent_hidden_states}
        next_value = pol.get_value(ob, hx, rollouts.masks[-1]).detach()
    rollouts.compute_returns(next_value, True, 0.99, 0.95, False)
    returns = rollouts.returns.numpy().copy()
    adv = rollouts.returns[:-1] - rollouts.value_preds[:-1]
    adv_n = ((adv - adv.mean()) / (adv.std() + 1e-5)).numpy().copy()
    # evaluate_actions on the first minibatch layout (envs 0..E/nmb-1 in order) for a forward-only check
    N = E // nmb
    sel = list(range(N))
    flat = lambda x: x[:, sel].reshape(T * N, *x.shape[2:])
    with torch.no_grad():
        ob_b = {k: flat(rollouts.obs[k][:-1]) for k in rollouts.obs}
        hx_b = {k: rollouts.recurrent_hidden_states[k][0, sel] for k in rollouts.recurrent_hidden_states}
        ev_values, ev_logp, ev_ent, _ = pol.evaluate_actions(ob_b, hx_b, flat(rollouts.masks[:-1]), flat(rollouts.actions))
    agent = ref_ppo.PPO(pol, 0.2, 2, nmb, 0.5, 0.0, lr=4e-5, eps=1e-5, max_grad_norm=0.5)
    torch.manual_seed(321)
    v_loss, a_loss, ent = agent.update(rollouts)
    sd_after = {k: v.detach().numpy().copy() for k, v in pol.state_dict().items()}
    checks = {k: np.array([float(np.sum(v.astype(np.float64))), float(np.sum(np.abs(v.astype(np.float64))))]) for k, v in sd_after.items()}
    out = dict(rewards=rewards, dones=dones, actions=np.array(actions_rec), logp=np.array(logp_rec), values=np.array(value_rec),
               next_value=next_value.numpy(), returns=returns, adv_norm=adv_n, masks=rollouts.masks.numpy().copy(),
               hxs_node=rollouts.recurrent_hidden_states["human_node_rnn"].numpy().copy(),
               ev_values=ev_values.numpy(), ev_logp=ev_logp.numpy(), ev_entropy=np.float32(ev_ent.item()),
               losses=np.array([v_loss, a_loss, ent], dtype=np.float64),
               meta=np.array(json.dumps(dict(env_name=env_name, E=E, H=H, D=D, T=T, nmb=nmb, act_seed=123, update_seed=321,
                                             ppo_epoch=2, shapes={k: list(v) for k, v in shapes.items()}))))
    for s in range(T + 1):
        for k, v in obs_seq[s].items():
            out["obs%d_%s" % (s, k)] = v
    for k, v in checks.items():
        out["chk_" + k] = v
    # every post-update tensor at <= 256 evenly spaced flat positions (np.linspace(0, n - 1, min(n, 256)).astype(int64)): the checksums
    # above average over millions of entries, these pin individual weights of EVERY layer
    for k, v in sd_after.items():
        flat = v.reshape(-1)
        out["smp_" + k] = flat[np.linspace(0, flat.size - 1, min(flat.size, 256)).astype(np.int64)].copy()
    # full post-update tensors for two small layers (tight check of Adam + clipping)
    out["after_dist.fc_mean.weight"] = sd_after["dist.fc_mean.weight"]
    out["after_base.critic_linear.weight"] = sd_after["base.critic_linear.weight"]
    out["after_base.robot_linear.0.weight"] = sd_after["base.robot_linear.0.weight"]
    path = os.path.join(HERE, "rollout_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print("rollout %-12s losses=%s -> %s (%.0f KB)" % (tag, out["losses"], os.path.basename(path), os.path.getsize(path) / 1024))


def init_case():
    """Parameter checksums of the reference Policy right after construction under torch.manual_seed(0)."""
    import torch
    from rl.networks.model import Policy
    out = {}
    for tag, env_name, H, D in (("varnum_h20", "CrowdSimVarNum-v0", 20, 2), ("pred_h20", "CrowdSimPred-v0", 20, 12)):
        args = ref_args(env_name, 16, 2, 30)
        ob_space, act_space = spaces(H, D)
        torch.manual_seed(0)
        pol = Policy(ob_space.spaces, act_space, base_kwargs=args, base="selfAttn_merge_srnn")
        for k, v in pol.state_dict().items():
            a = v.detach().numpy().astype(np.float64)
            out["%s/%s" % (tag, k)] = np.array([a.sum(), np.abs(a).sum(), float(a.ravel()[0]), float(a.ravel()[-1])])
    path = os.path.join(HERE, "policy_init.npz")
    np.savez_compressed(path, **out)
    print("policy init checksums -> %s (%.0f KB)" % (os.path.basename(path), os.path.getsize(path) / 1024))


def main():
    flags = set(sys.argv[1:])          # (install() rewrites sys.argv for the reference's config module)
    R.install()
    if "--variants-only" in flags:
        variant_case("varnum_h20_noattn", "CrowdSimVarNum-v0", 4, 20, 2, 3, False, True)
        variant_case("varnum_h20_unsorted", "CrowdSimVarNum-v0", 4, 20, 2, 3, True, False)
        variant_case("pred_h10_noattn_unsorted", "CrowdSimPred-v0", 3, 10, 12, 4, False, False)
        rollout_variant_case("varnum_e4_h8_t5_noattn_unsorted", "CrowdSimVarNum-v0", 4, 8, 2, 5, 2, False, False)
        rollout_variant_case("pred_e4_h10_t4_unsorted", "CrowdSimPred-v0", 4, 10, 12, 4, 2, True, False)
        return
    if "--rollouts-only" in flags:
        rollout_case("varnum_e4_h5_t6", "CrowdSimVarNum-v0", 4, 5, 2, 6, 2)
        rollout_case("pred_e4_h20_t5", "CrowdSimPred-v0", 4, 20, 12, 5, 2)
        return
    init_case()
    policy_case("varnum_e4_h20", "CrowdSimVarNum-v0", 4, 20, 2)
    policy_case("varnum_e1_h5", "CrowdSimVarNum-v0", 1, 5, 2)
    policy_case("pred_e4_h20", "CrowdSimPred-v0", 4, 20, 12)
    policy_case("varnum_e3_h50", "CrowdSimVarNum-v0", 3, 50, 2)
    rollout_case("varnum_e4_h5_t6", "CrowdSimVarNum-v0", 4, 5, 2, 6, 2)
    rollout_case("pred_e4_h20_t5", "CrowdSimPred-v0", 4, 20, 12, 5, 2)


if __name__ == "__main__":
    main()
