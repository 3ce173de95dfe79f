"""Generate tests/golden/learner_a2c_*.npz and learner_ppo_H64.npz from the REFERENCE's own A2CNetwork / PPONetwork
(marlbase/ac/model.py:21-352, imported unmodified with the four stubs of oracle/make_golden.py).
Runs only in the build container (needs /root/reference); the vectors travel, the reference does not.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_ac

Each file: actor / critic / target-critic blocks ([P][n], parameters() order), one rollout Batch per update in the
ac/train.py layout, the four metrics of update(), the gradient of the first update (recomputed without stepping),
and the blocks after each of 3 updates at env steps 0, 250, 400 (hard target copy at `step % 200 == 0`: updates 1, 3).
  learner_a2c_H64.npz       ia2c.yaml defaults (n_steps 5, entropy 0.001, value coef 0.5, no clipping), 2 agents x 15 obs
  learner_a2c_clip_H128.npz grad_clip 0.5 (clip over actor AND critic), n_steps 3, 3 agents x 18 obs, 128-128
  learner_ppo_H64.npz       ippo.yaml defaults (4 epochs, clip 0.2)
"""
import contextlib
import io
import os

import numpy as np
import torch

from .ac_update_port import synthetic_batch
from .make_golden import OUT, Box, Cfg, Discrete, flat_params, import_reference


def build(cls, P, D, A, H, seed, **over):
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
              standardise_returns=False, target_update_interval_or_tau=200, num_epochs=4, ppo_clip=0.2)
    cfg.update(over)
    centralised = bool(over.pop("centralised", False)) if "centralised" in over else False
    cfg.pop("centralised", None)
    net_cfg = dict(layers=list(cfg.pop("layers", [H, H])), parameter_sharing=False, use_orthogonal_init=True, use_rnn=bool(cfg.pop("use_rnn", False)))
    # actor.use_rnn / critic.use_rnn set separately (ac/model.py:45-97 passes each flag to its own family)
    actor_rnn, critic_rnn = cfg.pop("actor_rnn", None), cfg.pop("critic_rnn", None)
    a_cfg = dict(net_cfg, use_rnn=net_cfg["use_rnn"] if actor_rnn is None else bool(actor_rnn))
    if "actor_layers" in cfg:
        a_cfg["layers"] = list(cfg.pop("actor_layers"))
    c_cfg = dict(net_cfg, use_rnn=net_cfg["use_rnn"] if critic_rnn is None else bool(critic_rnn), centralised=centralised)
    if "critic_layers" in cfg:  # critic.layers is its own list (ac/model.py:45-97)
        c_cfg["layers"] = list(cfg.pop("critic_layers"))
    with contextlib.redirect_stdout(io.StringIO()):
        net = cls([Box(D)] * P, [Discrete(A)] * P, cfg, Cfg(a_cfg), Cfg(c_cfg), "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # non-zero biases, target != critic
        for p in list(net.actor.parameters()) + list(net.critic.parameters()):
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for p in net.target_critic.parameters():
            p.add_(0.08 * torch.randn(p.shape, generator=g))
    return net, cfg


def fixture(ref_ac_model, ref_ac_train, name, cls, P, D, H, N, seed, masked=False, **over):
    T, A = 25, 6
    net, cfg = build(cls, P, D, A, H, seed, **over)
    out = dict(P=P, T=T, N=N, D=D, A=A, H=H, actor_rnn=int(bool(over.get("actor_rnn", over.get("use_rnn", False)))),
               critic_rnn=int(bool(over.get("critic_rnn", over.get("use_rnn", False)))), state_dict_keys=np.array(list(net.state_dict().keys())),
               n_steps=cfg.n_steps, gamma=cfg.gamma, entropy_coef=cfg.entropy_coef,
               value_loss_coef=cfg.value_loss_coef, grad_clip=float(cfg.grad_clip or 0.0), num_epochs=cfg.num_epochs,
               ppo_clip=cfg.ppo_clip, actor0=flat_params(net.actor).numpy(), critic0=flat_params(net.critic).numpy(),
               target0=flat_params(net.target_critic).numpy())
    if "layers" in over:
        out["layers"] = np.array(over["layers"])
    if "critic_layers" in over:
        out["critic_layers"] = np.array(over["critic_layers"])
    if "actor_layers" in over:
        out["actor_layers"] = np.array(over["actor_layers"])
    steps = [0, 250, 400]
    batches = [synthetic_batch(P, T, N, D, A, seed=seed + 100 + i) for i in range(3)]
    if masked:  # batch.action_masks [T+1][N][P][A] (ac/train.py:53-63): random, never empty, the taken action allowed
        for i, b in enumerate(batches):
            g = torch.Generator().manual_seed(seed + 300 + i)
            m = (torch.rand(T + 1, N, P, A, generator=g) < 0.6).float()
            m[..., 0] = torch.maximum(m[..., 0], (m.sum(-1) == 0).float())
            m[:-1].scatter_(-1, b["actions"].unsqueeze(-1), 1.0)
            b["action_masks"] = m
    mk = lambda b: ref_ac_train.Batch(b["obss"], b["actions"], b["rewards"], b["dones"], b["filled"], b.get("action_masks"))  # noqa: E731
    if cls is ref_ac_model.A2CNetwork and not over.get("use_rnn") and not over.get("actor_rnn") and not over.get("critic_rnn"):  # gradient of the first update, via a throw-away copy stepped with lr = 0
        probe, _ = build(cls, P, D, A, H, seed, **dict(over, lr=0.0, grad_clip=False))  # noqa
        probe.update(mk(batches[0]), 1)
        out["actor_grad0"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in probe.actor.independent]).numpy()
        out["critic_grad0"] = torch.stack([torch.cat([p.grad.reshape(-1) for p in m.parameters()]) for m in probe.critic.independent]).numpy()
        with torch.no_grad():  # forward pieces for a forward-only check
            b = batches[0]
            nv, _ = probe.get_value(probe.split_obs(b["obss"]), None, target=True)
            out["next_value0"] = nv.numpy()
            done = b["dones"].float().unsqueeze(-1).repeat(1, 1, P)
            from marlbase.utils.utils import compute_nstep_returns
            out["returns0"] = compute_nstep_returns(b["rewards"], done, nv, cfg.n_steps, cfg.gamma).numpy()
    metrics = []
    for i, (b, st) in enumerate(zip(batches, steps)):
        m = net.update(mk(b), st)
        metrics.append([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]])
        out[f"actor{i + 1}"] = flat_params(net.actor).numpy()
        out[f"critic{i + 1}"] = flat_params(net.critic).numpy()
        out[f"target{i + 1}"] = flat_params(net.target_critic).numpy()
    out["metrics"] = np.array(metrics, np.float64)
    out["steps"] = np.array(steps)
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"batch{i}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "metrics", np.array(metrics).round(5).tolist())


def hetero_fixture(ref_ac_model, ref_ac_train, name, cls, obs_dims, act_dims, H, N, seed, **over):
    """agents with different observation / action sizes (MultiAgentIndependentNetwork builds each agent's network from its own sizes,
    utils/models.py:133-155; A2CNetwork splits the concatenated observation row by them, ac/model.py:115): the state_dict before and after
    each of 3 updates (tensor by tensor: the agents' blocks have different sizes), metrics, the batches in the reference's layout
    (obss [T+1, N, sum d_p]; actions of agent p below act_dims[p])"""
    T, P = 25, len(obs_dims)
    torch.manual_seed(seed)
    cfg = Cfg(optimizer="Adam", lr=3e-4, gamma=0.99, grad_clip=False, n_steps=5, entropy_coef=0.001, value_loss_coef=0.5,
              standardise_returns=False, target_update_interval_or_tau=200, num_epochs=4, ppo_clip=0.2)
    cfg.update(over)
    net_cfg = dict(layers=[H, H], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    with contextlib.redirect_stdout(io.StringIO()):
        net = cls([Box(d) for d in obs_dims], [Discrete(a) for a in act_dims], cfg, Cfg(net_cfg), Cfg(dict(net_cfg, centralised=False)), "cpu")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in list(net.actor.parameters()) + list(net.critic.parameters()):
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for p in net.target_critic.parameters():
            p.add_(0.08 * torch.randn(p.shape, generator=g))
    out = dict(P=P, T=T, N=N, H=H, obs_dims=np.array(obs_dims), act_dims=np.array(act_dims), state_dict_keys=np.array(list(net.state_dict().keys())),
               n_steps=cfg.n_steps, gamma=cfg.gamma, entropy_coef=cfg.entropy_coef, value_loss_coef=cfg.value_loss_coef, num_epochs=cfg.num_epochs,
               ppo_clip=cfg.ppo_clip)
    for k, v in net.state_dict().items():
        out[f"sd0/{k}"] = v.detach().numpy().copy()
    Dm, Am = max(obs_dims), max(act_dims)
    steps, metrics = [0, 250, 400], []
    for i, st in enumerate(steps):
        b = synthetic_batch(P, T, N, Dm, Am, seed=seed + 100 + i)  # the homogeneous generator at the widest sizes, cut down per agent
        obss = torch.cat([b["obss"][..., p * Dm:p * Dm + d] for p, d in enumerate(obs_dims)], dim=-1)
        acts = torch.stack([b["actions"][..., p] % a for p, a in enumerate(act_dims)], dim=-1)
        m = net.update(ref_ac_train.Batch(obss, acts, b["rewards"], b["dones"], b["filled"], None), st)
        metrics.append([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]])
        for k, v in (("obss", obss), ("actions", acts), ("rewards", b["rewards"]), ("dones", b["dones"]), ("filled", b["filled"])):
            out[f"batch{i}_{k}"] = v.numpy()
        for k, v in net.state_dict().items():
            out[f"sd{i + 1}/{k}"] = v.detach().numpy().copy()
    out["metrics"], out["steps"] = np.array(metrics, np.float64), np.array(steps)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "metrics", np.array(metrics).round(5).tolist())


def hetero(ram, rat):
    hetero_fixture(ram, rat, "learner_a2c_hetero_H64.npz", ram.A2CNetwork, obs_dims=[15, 18], act_dims=[6, 5], H=64, N=12, seed=2600)
    hetero_fixture(ram, rat, "learner_ppo_hetero_h48.npz", ram.PPONetwork, obs_dims=[12, 18, 15], act_dims=[4, 6, 6], H=48, N=9, seed=2700)


def stacked(ram, rat):
    """round 6: use_rnn with layers [h] * (L + 1) - nn.GRU(num_layers=L) in both families (utils/models.py:74-90)"""
    fixture(ram, rat, "learner_a2c_gru_L2_h24.npz", ram.A2CNetwork, P=2, D=15, H=24, N=12, seed=2100, use_rnn=True, layers=[24, 24, 24])
    fixture(ram, rat, "learner_mappo_gru_L3_p3_h40.npz", ram.PPONetwork, P=3, D=18, H=40, N=9, seed=2200, use_rnn=True, centralised=True, layers=[40, 40, 40, 40])
    # recurrent families of two depths: actor.layers [24, 24] (one GRU layer), critic.layers [24] * 4 (three)
    fixture(ram, rat, "learner_a2c_gru_L1_L3_h24.npz", ram.A2CNetwork, P=2, D=15, H=24, N=10, seed=2300, use_rnn=True, layers=[24, 24], critic_layers=[24, 24, 24, 24])
    # a stack next to a feed-forward family: recurrent actors [24] * 3 + feed-forward critics [24, 24]; feed-forward actors + recurrent critics [24] * 3
    fixture(ram, rat, "learner_a2c_rnn_actor_L2_ff_critic_h24.npz", ram.A2CNetwork, P=2, D=15, H=24, N=10, seed=2400, actor_rnn=True, critic_rnn=False,
            actor_layers=[24, 24, 24])
    fixture(ram, rat, "learner_ppo_ff_actor_rnn_critic_L2_h24.npz", ram.PPONetwork, P=2, D=15, H=24, N=10, seed=2500, actor_rnn=False, critic_rnn=True,
            critic_layers=[24, 24, 24])


if __name__ == "__main__":
    import sys

    torch.set_num_threads(1)
    import_reference()
    from marlbase.ac import model as ram
    from marlbase.ac import train as rat

    if "--stacked-only" in sys.argv:
        stacked(ram, rat)
        sys.exit(0)
    if "--hetero-only" in sys.argv:
        hetero(ram, rat)
        sys.exit(0)

    fixture(ram, rat, "learner_a2c_H64.npz", ram.A2CNetwork, P=2, D=15, H=64, N=12, seed=500)
    fixture(ram, rat, "learner_a2c_clip_H128.npz", ram.A2CNetwork, P=3, D=18, H=128, N=9, seed=600, grad_clip=0.5, n_steps=3)
    fixture(ram, rat, "learner_ppo_H64.npz", ram.PPONetwork, P=2, D=15, H=64, N=12, seed=700)
    # critic.centralised = True (maa2c.yaml / mappo.yaml): every critic reads all agents' observations
    fixture(ram, rat, "learner_maa2c_H64.npz", ram.A2CNetwork, P=2, D=15, H=64, N=12, seed=800, centralised=True)
    fixture(ram, rat, "learner_mappo_p3_H128.npz", ram.PPONetwork, P=3, D=18, H=128, N=9, seed=900, centralised=True)
    # batch.action_masks set (get_dist masks the logits, ac/model.py:135-145)
    fixture(ram, rat, "learner_a2c_masks_H64.npz", ram.A2CNetwork, P=2, D=15, H=64, N=12, seed=1300, masked=True)
    fixture(ram, rat, "learner_ppo_masks_H128.npz", ram.PPONetwork, P=2, D=15, H=128, N=10, seed=1400, masked=True)
    # recurrent actors and critics (use_rnn: RNNNetwork, utils/models.py:51-116)
    fixture(ram, rat, "learner_a2c_gru_H64.npz", ram.A2CNetwork, P=2, D=15, H=64, N=12, seed=1500, use_rnn=True)
    fixture(ram, rat, "learner_ppo_gru_H128.npz", ram.PPONetwork, P=2, D=15, H=128, N=10, seed=1600, use_rnn=True)
    fixture(ram, rat, "learner_maa2c_gru_H64.npz", ram.A2CNetwork, P=2, D=15, H=64, N=11, seed=1700, use_rnn=True, centralised=True)
    fixture(ram, rat, "learner_mappo_gru_p3_H64.npz", ram.PPONetwork, P=3, D=18, H=64, N=9, seed=1800, use_rnn=True, centralised=True)
    # actor.use_rnn != critic.use_rnn (round 6): recurrent actors next to feed-forward critics, and the reverse
    fixture(ram, rat, "learner_a2c_rnn_actor_ff_critic_H64.npz", ram.A2CNetwork, P=2, D=15, H=64, N=12, seed=1900, actor_rnn=True, critic_rnn=False)
    fixture(ram, rat, "learner_ppo_ff_actor_rnn_critic_H64.npz", ram.PPONetwork, P=2, D=15, H=64, N=10, seed=2000, actor_rnn=False, critic_rnn=True)
    stacked(ram, rat)
    hetero(ram, rat)
