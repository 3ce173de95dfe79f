"""Agents with different observation / action sizes in the actor-critic learners.  The reference builds each agent's network from its own
sizes (MultiAgentIndependentNetwork, marlbase/utils/models.py:133-155), splits the concatenated observation row by them and keeps one
Categorical per agent (marlbase/ac/model.py:115-145); its DQN family cannot train such agents (the learner stacks the agents' values,
marlbase/dqn/model.py:128, and the replay stacks their observations, marlbase/dqn/train.py:98).  Here they run on the kernels of
(max D, max A): zero input columns / output rows behind an agent's own, zero-padded rows, the missing actions masked as batch.action_masks
masks.  Goldens from the reference's own A2CNetwork / PPONetwork (oracle/make_golden_ac.py: hetero()): the padding argument itself against
them on the CPU (oracle/ac_update_port at the padded sizes), the driver classes on the GPU."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
FILES = ["learner_a2c_hetero_H64.npz", "learner_ppo_hetero_h48.npz"]
LAYERS = ("network.0", "network.2", "network.4")


def _load(name):
    g = dict(np.load(os.path.join(G, name)))
    return g, [int(x) for x in g["obs_dims"]], [int(x) for x in g["act_dims"]], int(g["H"])


def _padded_block(g, tag, family, p, d, a, D, A, H):
    """agent p's tensors of state_dict snapshot `tag` laid out for D inputs / A outputs, flat in parameters() order"""
    w1, b1, w2, b2, w3, b3 = (torch.tensor(g[f"{tag}/{family}.independent.{p}.{n}.{k}"]) for n in LAYERS for k in ("weight", "bias"))
    W1 = torch.zeros(H, D)
    W1[:, :d] = w1
    W3, B3 = torch.zeros(A, H), torch.zeros(A)
    W3[:a], B3[:a] = w3, b3
    return torch.cat([W1.reshape(-1), b1, w2.reshape(-1), b2, W3.reshape(-1), B3])


def _padded_batch(g, i, obs_dims, act_dims):
    D, A, P = max(obs_dims), max(act_dims), len(obs_dims)
    obss = torch.tensor(g[f"batch{i}_obss"])
    wide, o = torch.zeros(*obss.shape[:-1], P * D), 0
    for p, d in enumerate(obs_dims):
        wide[..., p * D:p * D + d] = obss[..., o:o + d]
        o += d
    own = (torch.arange(A)[None, :] < torch.tensor(act_dims)[:, None]).float()
    b = {k: torch.tensor(g[f"batch{i}_{k}"]) for k in ("actions", "rewards", "dones", "filled")}
    b["obss"] = wide
    b["action_masks"] = own.expand(obss.shape[0], obss.shape[1], P, A).contiguous()
    return b


@pytest.mark.parametrize("name", FILES)
def test_padding_and_masks_reproduce_the_reference_on_the_port(name):
    """the equivalence the HIP path rests on, in the oracle: blocks and rows padded to (max D, max A) + the missing actions masked give the
    reference's metrics and, in the agents' own entries, its parameters after every update; the padding never moves"""
    from oracle import ac_update_port as ap
    from oracle import dqn_port as dp

    g, obs_dims, act_dims, H = _load(name)
    P, D, A = len(obs_dims), max(obs_dims), max(act_dims)
    actor = torch.stack([_padded_block(g, "sd0", "actor", p, obs_dims[p], act_dims[p], D, A, H) for p in range(P)])
    critic = torch.stack([_padded_block(g, "sd0", "critic", p, obs_dims[p], 1, D, 1, H) for p in range(P)])
    target = torch.stack([_padded_block(g, "sd0", "target_critic", p, obs_dims[p], 1, D, 1, H) for p in range(P)])
    lr = ap.Learner(actor, critic, D, H, A, gamma=float(g["gamma"]), n_steps=int(g["n_steps"]), entropy_coef=float(g["entropy_coef"]),
                    value_loss_coef=float(g["value_loss_coef"]), num_epochs=int(g["num_epochs"]) if "ppo" in name else 0, ppo_clip=float(g["ppo_clip"]))
    lr.target = target
    for i in range(3):
        m = lr.update(_padded_batch(g, i, obs_dims, act_dims), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=2e-5, atol=2e-6)
        for fam, blocks, outs in (("actor", lr.actor().detach(), act_dims), ("critic", lr.critic().detach(), [1] * P)):
            for p in range(P):
                want = _padded_block(g, f"sd{i + 1}", fam, p, obs_dims[p], outs[p], D, A if fam == "actor" else 1, H)
                np.testing.assert_allclose(blocks[p].numpy(), want.numpy(), rtol=0, atol=3e-6)  # (incl. the padding: zeros in `want`)


def test_init_draws_of_agents_of_different_sizes_are_the_references():
    """_init_blocks_io_padded consumes torch's RNG as FCNetwork does agent by agent at the agents' own sizes (utils/models.py:34-48)"""
    from codebase_amd.ac.model import _init_blocks_io_padded
    from codebase_amd.dqn.model import _fc

    torch.manual_seed(3)
    got = _init_blocks_io_padded([12, 18], [48, 48], [4, 6], True, 18, 6)
    torch.manual_seed(3)
    for p, (d, a) in enumerate(((12, 4), (18, 6))):
        lins = _fc([d, 48, 48, a], True)
        o = 0
        W1 = got[p, o:o + 48 * 18].view(48, 18)
        assert torch.equal(W1[:, :d], lins[0].weight.detach()) and float(W1[:, d:].abs().max() if d < 18 else 0.0) == 0.0
        o += 48 * 18 + 48 + 48 * 48 + 48
        W3 = got[p, o:o + 6 * 48].view(6, 48)
        assert torch.equal(W3[:a], lins[2].weight.detach()) and float(W3[a:].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_hip_actor_critic_with_agents_of_different_sizes_matches_reference(name):
    from collections import namedtuple

    from codebase_amd.ac.model import A2CNetwork, PPONetwork
    from codebase_amd.spaces import Box, Discrete, Tuple

    Batch = namedtuple("Batch", ["obss", "actions", "rewards", "dones", "filled", "action_masks"])
    g, obs_dims, act_dims, H = _load(name)
    P, ppo = len(obs_dims), "ppo" in name
    cfg = dict(optimizer="Adam", lr=3e-4, gamma=float(g["gamma"]), grad_clip=False, n_steps=int(g["n_steps"]), entropy_coef=float(g["entropy_coef"]),
               value_loss_coef=float(g["value_loss_coef"]), standardise_returns=False, target_update_interval_or_tau=200,
               num_epochs=int(g["num_epochs"]), ppo_clip=float(g["ppo_clip"]))
    net_cfg = dict(layers=[H, H], parameter_sharing=False, use_orthogonal_init=True, use_rnn=False)
    net = (PPONetwork if ppo else A2CNetwork)(Tuple([Box(-1, 8, (d,)) for d in obs_dims]), Tuple([Discrete(a) for a in act_dims]), cfg, dict(net_cfg),
                                             dict(net_cfg, centralised=False), "cuda")
    assert net.hetero and net.spec.obs_dim == max(obs_dims) and net.spec.n_actions == max(act_dims)
    sd = net.state_dict()
    keys = [str(k) for k in g["state_dict_keys"]]
    assert list(sd.keys()) == keys and all(tuple(sd[k].shape) == g[f"sd0/{k}"].shape for k in keys)  # the reference's keys AND shapes, agent by agent
    net.load_state_dict({k: torch.tensor(g[f"sd0/{k}"]) for k in keys})
    pad_a, pad_c = torch.ones_like(net.actor_params, dtype=torch.bool), torch.ones_like(net.critic_params, dtype=torch.bool)
    views = net._views()
    for (blocks, mask, fam) in ((net.actor_params, pad_a, "actor"), (net.critic_params, pad_c, "critic")):
        for k, v in views.items():  # mark the live entries through the views' own storage offsets
            if k.startswith(fam + "."):
                p = int(k.split(".")[2])
                idx = torch.zeros_like(blocks[p], dtype=torch.bool)
                idx.as_strided(v.shape, v.stride(), v.storage_offset() - blocks[p].storage_offset()).fill_(True)
                mask[p] &= ~idx
    assert int(pad_a.sum()) > 0 and float(net.actor_params[pad_a].abs().max()) == 0.0 and float(net.critic_params[pad_c].abs().max()) == 0.0
    # acting / values with per-agent rows of the agents' own widths; an agent never draws an action it does not have
    N = 64
    obs = [torch.rand(N, d) for d in obs_dims]
    for _ in range(8):
        acts, _ = net.act(obs, net.init_actor_hiddens(N))
        assert acts.shape == (P, N, 1) and all(int(acts[p].max()) < act_dims[p] for p in range(P))
    v, _ = net.get_value(obs, net.init_critic_hiddens(N))
    assert v.shape == (N, P)
    for i in range(3):
        b = Batch(*(torch.tensor(g[f"batch{i}_{k}"]).cuda() for k in ("obss", "actions", "rewards", "dones", "filled")), None)
        m = net.update(b._replace(dones=b.dones.float()), int(g["steps"][i]))
        np.testing.assert_allclose([m["loss"], m["actor_loss"], m["value_loss"], m["entropy"]], g["metrics"][i], rtol=1e-4, atol=1e-5)
        sd = net.state_dict()
        worst = max(float(np.abs(sd[k].cpu().numpy() - g[f"sd{i + 1}/{k}"]).max()) for k in keys)
        off = sum(int((np.abs(sd[k].cpu().numpy() - g[f"sd{i + 1}/{k}"]) > 5e-6).sum()) for k in keys)
        total = sum(g[f"sd{i + 1}/{k}"].size for k in keys)
        assert worst <= 5e-5 and off <= 1e-4 * total, (i, worst, off)
        assert float(net.actor_params[pad_a].abs().max()) == 0.0 and float(net.critic_params[pad_c].abs().max()) == 0.0  # the padding never moves
    with pytest.raises(NotImplementedError):  # the combinations that stay out: a centralised critic, sharing, use_rnn
        A2CNetwork(Tuple([Box(-1, 8, (d,)) for d in obs_dims]), Tuple([Discrete(a) for a in act_dims]), cfg, dict(net_cfg), dict(net_cfg, centralised=True), "cuda")
