"""Host-side logic of `algorithm.model.layers` handling (codebase_amd/dqn/model.py: compiled_width, is_wide, pad_blocks, block_views),
of `algorithm.optimizer` (hip.optimizer_id) and of the rollout infos (ac/train.py EpisodeInfo) - no GPU.  The claim behind the zero
padding ("a padded unit has zero weights and bias -> activation 0, every gradient into the padding is a product with one of those
zeros") is checked here with the oracle port: the padded block evaluated at the padded widths equals the live block at the true
widths, and a loss's gradient is exactly zero on the padding."""
import pytest
import torch

from codebase_amd import hip as h
from codebase_amd.dqn import model as M
from oracle import dqn_port as dp


def test_which_kernel_family_a_layer_list_runs_on():
    assert M.compiled_width([64, 64]) == 64 and M.compiled_width([32, 48]) == 64 and M.compiled_width([1, 1]) == 64
    assert M.compiled_width([128, 128]) == 128 and M.compiled_width([65, 8]) == 128 and M.compiled_width([128, 40]) == 128
    for hidden in ([64, 64], [128, 40], [32, 32]):
        assert not M.is_wide(hidden)
    # the GEMM path: widths rounded up to 16, two-layer lists at least one tile wider than the fused kernels' 128
    assert M.compiled_width([256, 256]) == 256 and M.compiled_width([200, 96]) == 208 and M.compiled_width([129, 5]) == 144
    assert M.compiled_width([64]) == 64 and M.compiled_width([100, 50, 30, 20]) == 112 and M.compiled_width([8, 8, 8]) == 16
    for hidden in ([256, 256], [129, 5], [64], [64, 64, 64], [100, 50, 30, 20]):
        assert M.is_wide(hidden)
    assert M.compiled_width([64] * 5) == 64 and M.is_wide([64] * 5) and M.recurrent_width([40, 40]) == (40, 64) and M.recurrent_width([100, 100]) == (100, 128)
    for hidden in ([], [64] * 17, [2048, 2048], [0, 64], [1025]):
        with pytest.raises(NotImplementedError):
            M.compiled_width(hidden)


@pytest.mark.parametrize("hidden", [(48, 24), (128, 40), (200, 96), (96,), (100, 50, 30, 20)])
def test_zero_padding_is_exact_and_takes_no_gradient(hidden):
    P, D, A = 3, 15, 6
    H = M.compiled_width(list(hidden))
    live = dp.init_params(P, D, hidden, A, seed=4) + 0.02
    padded = M.pad_blocks(live, D, list(hidden), A, H)
    full = tuple([H] * len(hidden))
    assert padded.shape == (P, dp.nparams(D, full, A))
    assert int((padded != 0).sum()) == int((live != 0).sum())  # nothing but the live entries
    # block_views of the padded row are the live tensors, in parameters() order and with the reference's key names
    names = [n for n, _ in M.block_views(padded[0], D, list(hidden), A, H)]
    assert names == [f"network.{2 * k}.{s}" for k in range(len(hidden) + 1) for s in ("weight", "bias")]
    for p in range(P):
        for (_, v), t in zip(M.block_views(padded[p], D, list(hidden), A, H), dp.split(live[p], D, hidden, A)):
            assert torch.equal(v, t)
    x = torch.randn(P, 50, D, generator=torch.Generator().manual_seed(1))
    pr = padded.clone().requires_grad_(True)
    q_pad = dp.q_values(pr, x, D, full, A)
    q_live = dp.q_values(live, x, D, hidden, A)
    torch.testing.assert_close(q_pad.detach(), q_live, rtol=0, atol=2e-6)
    (q_pad ** 2).sum().backward()
    mask = torch.ones_like(padded, dtype=torch.bool)
    for p in range(P):
        row = torch.zeros(padded.shape[1], dtype=torch.bool)
        for _, v in M.block_views(row, D, list(hidden), A, H):
            v.fill_(True)
        mask[p] = row
    assert float(pr.grad[~mask].abs().sum()) == 0.0 and float(pr.grad[mask].abs().sum()) > 0.0


def test_equal_widths_are_not_copied():
    flat = dp.init_params(2, 15, 64, 6, seed=0)
    assert M.pad_blocks(flat, 15, [64, 64], 6, 64) is flat


def test_optimizer_names():
    assert h.optimizer_id("Adam") == 0
    assert {h.optimizer_id(n) for n in ("Adam", "SGD", "RMSprop", "AdamW")} == set(range(4))
    assert h.optimizer_id(torch.optim.SGD) == h.optimizer_id("SGD")  # the class itself, as getattr(optim, cfg.optimizer) yields
    for bad in ("Adagrad", "LBFGS", torch.optim.Adamax):
        with pytest.raises(NotImplementedError):
            h.optimizer_id(bad)


def test_episode_info_is_a_dict_with_an_env_tag():
    from codebase_amd.ac.train import EpisodeInfo
    from codebase_amd.utils.loggers import squash_info

    a, b = EpisodeInfo(episode_returns=[1.0, 2.0], episode_length=5), EpisodeInfo(episode_returns=[3.0, 4.0], episode_length=7)
    a.env, b.env = 3, 9
    assert isinstance(a, dict) and "env" not in a and a.env == 3 and EpisodeInfo().env == -1
    out = squash_info([a, b])  # the tag never reaches the averaged keys
    assert set(out) == {"mean_episode_returns", "std_episode_returns", "mean_episode_length", "std_episode_length"}
    assert out["mean_episode_returns"] == 5.0 and out["mean_episode_length"] == 6.0
