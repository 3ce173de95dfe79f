"""CPU: libmarlhip.so loads and exports exactly the entry points include/marlhip.h declares; the
ctypes prototypes cover all of them; argument validation fails loudly without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "marlhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(marlhip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from codebase_amd import _lib

    syms = header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in marlhip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == syms, "ctypes prototypes and header disagree"
    assert _lib.lib.marlhip_version() == 219


def test_validation_errors_are_loud_and_need_no_gpu():
    from codebase_amd import _lib

    cfg = _lib.LbfConfig(n_envs=4, n_agents=7, n_food=9, rows=8, cols=8, sight=8, max_episode_steps=50, time_limit=25,
                         min_player_level=1, max_player_level=2, min_food_level=1, normalize_reward=1)
    assert _lib.lib.marlhip_lbf_state_stride(ctypes.byref(cfg)) < 0
    assert "no LBF kernel for 7p-9f" in _lib.last_error()
    with pytest.raises(_lib.MarlHipError):
        _lib.check(_lib.lib.marlhip_lbf_state_stride(ctypes.byref(cfg)), "stride")
    cfg.n_agents, cfg.n_food = 2, 3
    assert _lib.lib.marlhip_lbf_state_stride(ctypes.byref(cfg)) == 20
    assert _lib.lib.marlhip_lbf_obs_dim(ctypes.byref(cfg)) == 15
    s = _lib.NetShape(2, 15, 64, 6)
    assert _lib.lib.marlhip_net_nparams(ctypes.byref(s)) == 64 * 15 + 64 + 64 * 64 + 64 + 6 * 64 + 6
    s = _lib.NetShape(2, 15, 96, 6)
    assert _lib.lib.marlhip_net_nparams(ctypes.byref(s)) < 0
    assert _lib.lib.marlhip_dqn_workspace_bytes(ctypes.byref(_lib.NetShape(2, 15, 64, 6)), 25, 32) > 0
    # forward-only entry points take their weight-pack scratch from the caller: the size query needs no GPU
    n64, n128 = (_lib.lib.marlhip_forward_workspace_bytes(ctypes.byref(_lib.NetShape(2, 15, h, 6))) for h in (64, 128))
    assert 2 * 2 * 6288 * 4 <= n64 < n128
    assert _lib.lib.marlhip_forward_workspace_bytes(ctypes.byref(_lib.NetShape(0, 15, 64, 6))) < 0


def test_product_path_refuses_to_run_without_a_gpu():
    import torch

    from codebase_amd import hip as h

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(h.MarlHipError):
        h.BatchedForaging(h.lbf_config("lbforaging:Foraging-8x8-2p-3f-v3", 4, 25))
    with pytest.raises(h.MarlHipError):
        h.DeviceReplay(16, 2, 15, 25)


def test_product_code_never_imports_the_oracle():
    """codebase_amd/ and scripts/ never touch oracle/ (tools that use it as a checker / CPU baseline live under tests/tools/);
    bench.py only inside its cpu_baseline leg, __graft_entry__ only inside smoke()"""
    for sub in ("codebase_amd", "scripts"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{sub}/{f} imports the oracle"
    for f, fn in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        txt = open(os.path.join(ROOT, f)).read()
        for m in re.finditer(r"^(\s*)(from|import)\s+oracle\b", txt, flags=re.M):
            assert len(m.group(1)) > 0, f"{f}: module-level oracle import"
            head = txt[:m.start()]
            last_def = re.findall(r"^def (\w+)\(", head, flags=re.M)[-1]
            line = txt[m.start():txt.index("\n", m.start())]
            if f == "__graft_entry__.py" and last_def == "build" and line.strip() == "from oracle import make_ref":
                continue  # build() BUILDS the checker (oracle/_ref, the reference's sources for the cpu_baseline leg); it runs nothing of it
            assert fn in last_def, f"{f}: oracle imported inside {last_def}(), expected only inside *{fn}*()"


def test_integration_md_stub_structs_match_the_binding():
    """the ctypes stub printed in INTEGRATION.md declares the same struct layouts as codebase_amd/_lib.py (and so include/marlhip.h)"""
    from codebase_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# marlbase/utils/marlhip_stub.py\n(.*?)```", md, re.S).group(1)
    # the two struct classes only: no library load, no torch
    classes = "import ctypes\n" + "\n".join(m.group(0) for m in re.finditer(r"class \w+\(ctypes\.Structure\):.*?(?=\nclass |\nlib\.)", code, re.S))
    ns = {}
    exec(classes, ns)
    for stub, real in ((ns["NetShape"], _lib.NetShape), (ns["Batch"], _lib.BatchStruct)):
        assert ctypes.sizeof(stub) == ctypes.sizeof(real)
        assert [(n, ctypes.sizeof(t)) for n, t in stub._fields_] == [(n, ctypes.sizeof(t)) for n, t in real._fields_]
