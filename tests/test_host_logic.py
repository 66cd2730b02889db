"""CPU-side tests: option parsing against the spec frozen from the reference parser, state-dict key
compatibility, the C ABI (library loads, exports every symbol include/sqd.h declares, rejects CPU
tensors), the synthetic data schema, network wiring against the oracle, and the multi-process gradient
reducer on gloo (world_size 2)."""
import ast
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PRODUCT, REPO, tt


# ------------------------------------------------------------------------------------------ options
def test_options_match_reference_spec(golden):
    from options import MonodepthOptions
    g = golden("g00_options_spec")
    parser = MonodepthOptions().parser
    mine = {a.option_strings[0]: a for a in parser._actions if a.option_strings}
    for row in g["rows"]:
        flag, kind, typ, default, nargs, choices = str(row).split("|")
        if flag == "-h":
            continue
        assert flag in mine, flag
        a = mine[flag]
        assert type(a).__name__ == kind, flag
        want_default = ast.literal_eval(default)
        if flag == "--log_dir":
            want_default = os.path.join(os.path.expanduser("~"), "tmp")
        assert a.default == want_default, (flag, a.default, want_default)
        assert repr(a.nargs) == nargs and repr(a.choices) == choices, flag
        if kind == "_StoreAction":
            assert getattr(a.type, "__name__", str(a.type)) == typ, flag


def test_reference_args_files_parse_identically(golden):
    """Every args file of the reference parses to the same namespace (or fails the same way)."""
    from options import MonodepthOptions
    g = golden("g00_options_spec")
    checked = 0
    for name, parsed, text in zip(g["files"], g["parsed"], g["tokens"]):
        text = str(text)
        checked += 1
        if str(parsed) == "ARGPARSE_ERROR":
            with pytest.raises(SystemExit):
                MonodepthOptions().parser.parse_args(text.split())
            continue
        ns = vars(MonodepthOptions().parser.parse_args(text.split()))
        for k, v in ast.literal_eval(str(parsed)):
            assert ns[k] == v, (name, k, ns[k], v)
    assert checked >= 30


def test_unknown_reference_flag_errors_like_reference():
    from options import MonodepthOptions
    with pytest.raises(SystemExit):        # --model_type is not an options.py flag in the reference either
        MonodepthOptions().parser.parse_args(["--model_type", "hrb5"])
    assert MonodepthOptions().parser.parse_args([]).png == ".png"     # store_true with a truthy default (App. B-5)


# ------------------------------------------------------------------------------- state-dict surface
def test_state_dict_keys_match_reference(golden):
    import networks
    g = golden("g00_state_dict_keys")
    mods = {"encoder_res50": networks.ResnetEncoderDecoder(num_layers=50, num_features=256, model_dim=32),
            "encoder_res18": networks.LiteResnetEncoderDecoder(model_dim=32),
            "depth": networks.Depth_Decoder_QueryTr(in_channels=32, patch_size=16, dim_out=64, embedding_dim=32,
                                                    query_nums=64, num_heads=4),
            "pose": networks.PoseCNN(2)}
    for n, m in mods.items():
        mine = ["%s %s" % (k, tuple(v.shape)) for k, v in m.state_dict().items()]
        assert mine == list(g[n]), n
    assert networks.Lite_Depth_Decoder_QueryTr(in_channels=32, embedding_dim=32).transformer_encoder.layers[0].linear1.out_features == 512


def test_networks_match_oracle_on_cpu():
    """Same weights -> same outputs as the oracle restatement: module wiring and state-dict layout.  The product's operators
    refuse host tensors; the test patches plain-torch definitions of them (tests/host_ops.py) into the dispatch module."""
    import host_ops
    with host_ops.patched():
        _networks_match_oracle_on_cpu()


def test_product_network_ops_refuse_cpu_tensors():
    import networks
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        networks.PoseCNN(2)(torch.rand(1, 6, 64, 96))


def _networks_match_oracle_on_cpu():
    import networks
    from oracle import torch_ref as O
    from param_fill import fill_params
    torch.manual_seed(0)
    x = torch.rand(2, 3, 64, 96)
    for mine, ref in ((networks.LiteResnetEncoderDecoder(model_dim=16), O.LiteResnetEncoderDecoder(model_dim=16)),
                      (networks.ResnetEncoderDecoder(50, 64, 16), O.ResnetEncoderDecoder(50, 64, 16))):
        fill_params(ref, 5)
        mine.load_state_dict(ref.state_dict())
        mine.train(); ref.train()
        np.testing.assert_allclose(mine(x).detach().numpy(), ref(x).detach().numpy(), rtol=1e-4, atol=1e-5)
    feat = torch.randn(2, 16, 32, 48)
    mine = networks.Lite_Depth_Decoder_QueryTr(in_channels=16, embedding_dim=16, patch_size=8, query_nums=12, dim_out=24, max_val=80.0)
    ref = O.QueryTrDecoder(16, 16, 8, 4, 12, 24, max_val=80.0, dim_feedforward=512)
    fill_params(ref, 6)
    mine.load_state_dict(ref.state_dict())
    mine.eval(); ref.eval()
    np.testing.assert_allclose(mine(feat)[("disp", 0)].detach().numpy(), ref(feat)[("disp", 0)].detach().numpy(), rtol=1e-4, atol=1e-4)
    pm, pr = networks.PoseCNN(2), O.PoseCNN(2)
    fill_params(pr, 7)
    pm.load_state_dict(pr.state_dict())
    a, b = pm(torch.rand(2, 6, 64, 96)), pr(torch.rand(2, 6, 64, 96).mul(0) + 0.5)
    assert a[0].shape == b[0].shape == (2, 1, 1, 3)


# --------------------------------------------------------------------------------------------- ABI
def test_abi_exports_every_declared_symbol():
    from sqd import lib
    if lib.needs_build():
        lib.build()
    L = lib.lib()
    header = open(os.path.join(REPO, "include", "sqd.h")).read()
    declared = set(re.findall(r"\b(sqd_[a-z0-9_]+)\s*\(", header))
    assert declared, "no entry points parsed from include/sqd.h"
    for name in declared:
        assert hasattr(L, name), "libsqd.so lacks %s declared in include/sqd.h" % name
    assert set(lib.exported_symbols()) == declared, set(lib.exported_symbols()) ^ declared
    assert L.sqd_abi_version() == lib.ABI_VERSION == 3


def test_product_ops_refuse_cpu_tensors():
    from sqd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.identity_fwd(torch.zeros(1, 3, 16, 16), [torch.zeros(1, 3, 16, 16)] * 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pose_mats_fwd(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), [1, 0], torch.eye(4)[None])


def test_product_never_imports_oracle():
    for root, _, files in os.walk(PRODUCT):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(root, f)


# ------------------------------------------------------------------------------------------- data
def test_synthetic_batch_schema():
    from datasets import synthetic_batch
    b = synthetic_batch(2, 64, 96, with_gt=True)
    for f in (0, -1, 1):
        assert b[("color", f, 0)].shape == (2, 3, 64, 96) and b[("color_aug", f, 0)].shape == (2, 3, 64, 96)
        assert 0 <= float(b[("color", f, 0)].min()) and float(b[("color", f, 0)].max()) <= 1
    K = b[("K", 0)][0]
    assert abs(float(K[0, 0]) - 0.58 * 96) < 1e-4 and abs(float(K[1, 1]) - 1.92 * 64) < 1e-4
    np.testing.assert_allclose((b[("K", 0)][0] @ b[("inv_K", 0)][0]).numpy(), np.eye(4), atol=1e-4)
    assert b["depth_gt"].shape == (2, 1, 375, 1242) and float(b["depth_gt"].min()) > 0


# -------------------------------------------------------------------------------- multi-process DDP
def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, PRODUCT)
    from sqd import ddp
    ddp.init_from_env("gloo")
    torch.manual_seed(100 + rank)                      # different initial weights per rank on purpose
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
    unused = torch.nn.Linear(4, 4)                     # never receives a gradient (like the trunk's fc)
    params = list(model.parameters()) + list(unused.parameters())
    red = ddp.GradBucketReducer(params, bucket_mb=0.0005)     # tiny buckets -> several buckets
    red.broadcast_parameters([model, unused])
    torch.manual_seed(7)
    data = torch.randn(4, 3, 8, 8)
    shard = data[rank * 2:(rank + 1) * 2]
    opt = torch.optim.SGD(params, lr=0.1)
    hist = []
    for step in range(3):
        red.zero_grad()
        model(shard).square().mean().backward()
        red.finish()
        hist.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone())
        opt.step()
    q.put((rank, [h.numpy() for h in hist], [p.detach().numpy() for p in model.parameters()], len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_grad_bucket_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    (r0, g0, w0, nb0), (r1, g1, w1, nb1) = res
    assert nb0 == nb1 and nb0 >= 2
    for a, b in zip(g0, g1):
        np.testing.assert_allclose(a, b, rtol=0, atol=0)            # both ranks hold the same averaged gradient
    for a, b in zip(w0, w1):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)
    # single-process reference: full batch of 4 == mean of the two shards of 2
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
    torch.manual_seed(7)
    data = torch.randn(4, 3, 8, 8)
    (0.5 * (model(data[:2]).square().mean() + model(data[2:]).square().mean())).backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    np.testing.assert_allclose(g0[0], want, rtol=1e-5, atol=1e-7)


def _trainer_params_worker(rank, world, port, q):
    """The reducer on the Trainer's own parameter set: ResNet-18 encoder-decoder (with the never-used torchvision `fc`), the QTR
    head and PoseCNN, filters channels-last as the Trainer keeps them.  Gradients come from a CPU autograd pass over a
    surrogate loss (the kernels of the hot path need the GPU; the bucket logic does not)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, PRODUCT)
    import networks
    from sqd import ddp
    ddp.init_from_env("gloo")
    torch.manual_seed(10 + rank)
    models = [networks.LiteResnetEncoderDecoder(model_dim=16),
              networks.Lite_Depth_Decoder_QueryTr(in_channels=16, patch_size=8, dim_out=24, embedding_dim=16, query_nums=12, num_heads=4,
                                                  min_val=0.001, max_val=80.0),
              networks.PoseCNN(2)]
    for m in models:
        m.to(memory_format=torch.channels_last)
    params = [p for m in models for p in m.parameters()]
    names = [n for m in models for n, _ in m.named_parameters()]
    red = ddp.GradBucketReducer(params, bucket_mb=2.0)
    red.broadcast_parameters(models)
    unused = {i for i, n in enumerate(names) if n.startswith("encoder.encoder.fc.")}
    assert len(unused) == 2
    coef = [torch.full_like(p, float(rank + 1) * (1 + i % 3)) for i, p in enumerate(params)]
    hist = []
    for step in range(3):
        red.zero_grad()
        loss = sum((p * c).sum() for i, (p, c) in enumerate(zip(params, coef)) if i not in unused) * (step + 1)
        loss.backward()
        red.finish()
        hist.append([None if p.grad is None else p.grad.clone() for p in params])
    # a step in which some filters' gradients are DEFERRED (round 4: handed to the parameter directly by Conv2d.backward and announced
    # through nnkernels.DEFERRED_GRAD_HOOK = reducer.on_deferred_grad instead of passing through AccumulateGrad): every fourth 4-D
    # parameter is detached from the autograd loss and gets its gradient assigned by hand, in the middle of the backward pass' hooks
    deferred = [i for i, p in enumerate(params) if p.dim() == 4 and i not in unused][::4]
    red.zero_grad()
    loss = sum((p * c).sum() for i, (p, c) in enumerate(zip(params, coef)) if i not in unused and i not in deferred) * 4.0
    for i in deferred[: len(deferred) // 2]:                     # (some before, some after the autograd pass: arrival order is free)
        params[i].grad = (coef[i] * 4.0).contiguous(memory_format=torch.channels_last)
        red.on_deferred_grad(params[i])
    loss.backward()
    for i in deferred[len(deferred) // 2:]:
        params[i].grad = (coef[i] * 4.0).contiguous(memory_format=torch.channels_last)
        red.on_deferred_grad(params[i].detach().requires_grad_())        # (found by storage when it is not the parameter object itself)
    red.finish()
    hist.append([None if p.grad is None else p.grad.clone() for p in params])
    # a step in which a few bucketed parameters legitimately receive NO gradient (a frozen branch; the same ones on every rank): their
    # buckets are exchanged with zeros in their place — the used parameters of those buckets still get the averaged gradient
    skipped = [i for i in range(len(params)) if i not in unused][3::11]
    red.zero_grad()
    loss = sum((p * c).sum() for i, (p, c) in enumerate(zip(params, coef)) if i not in unused and i not in skipped) * 5.0
    loss.backward()
    red.finish()
    hist.append([None if p.grad is None else p.grad.clone() for p in params])
    zero_filled = all(float(hist[-1][i].abs().max()) == 0.0 for i in skipped)
    conv4d = next(i for i, p in enumerate(params) if p.dim() == 4 and p.shape[2] == 3 and p.shape[1] > 1)
    info = {"nbuckets": len(red.buckets), "ndeferred": len(deferred), "in_buckets": sum(len(b) for b in red.buckets), "nparams": len(params),
            "fc_grad_none": all(hist[-1][i] is None for i in unused),
            "view_strides_ok": params[conv4d].grad.stride() == params[conv4d].stride() and not params[conv4d].is_contiguous(),
            "is_view": params[conv4d].grad._base is not None, "zero_filled": zero_filled, "nskipped": len(skipped),
            "w0": float(params[0].detach().double().sum())}
    vals = [[None if g is None else float(g.double().mean()) for g in h] for h in hist]
    q.put((rank, info, vals))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_on_trainer_parameter_set_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_params_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    (_, i0, v0), (_, i1, v1) = res
    assert i0 == i1 or {k: v for k, v in i0.items() if k != "w0"} == {k: v for k, v in i1.items() if k != "w0"}
    assert i0["w0"] == i1["w0"]                                   # rank 0's weights were broadcast
    assert i0["nbuckets"] >= 2 and i0["in_buckets"] == i0["nparams"] - 2 and i0["fc_grad_none"]     # the unused fc stays out of the buckets
    assert i0["view_strides_ok"] and i0["is_view"]                # channels-last filters keep their layout inside the bucket
    assert i0["ndeferred"] >= 4 and len(v0) == 5                  # (the fourth step: deferred gradients announced by hand)
    assert i0["nskipped"] >= 3 and i0["zero_filled"]              # (the fifth: parameters without a gradient exchanged as zeros)
    # the used parameters of the fifth step: mean of (rank + 1) * (1 + i % 3) * 5 over the two ranks = 7.5 * (1 + i % 3)
    used5 = [(i, m) for i, m in enumerate(v0[4]) if m is not None and m != 0.0]
    assert len(used5) > 20 and all(abs(m - 7.5 * (1 + i % 3)) < 1e-4 for i, m in used5), used5[:5]
    for step, (a, b) in enumerate(zip(v0, v1)):
        assert a == b                                             # both ranks hold the same averaged gradients
        for i, g in enumerate(a):
            if g is not None and not (step == 4 and g == 0.0):   # mean of rank coefficients 1x and 2x = 1.5x, times the step factor (fifth step: zeros for the skipped parameters)
                assert abs(g - 1.5 * (1 + i % 3) * (step + 1)) < 1e-5, (step, i, g)


_TORCHRUN_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(product)r)
import torch
from sqd import ddp
rank, world, local = ddp.init_from_env("gloo")          # the launcher's environment: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
assert world == 2 and rank == int(os.environ["RANK"]) and type(ddp.COMM).__name__ == "GlooComm"
torch.manual_seed(rank)
model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
red = ddp.GradBucketReducer(list(model.parameters()), bucket_mb=0.0002)
red.broadcast_parameters([model])
w0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
for step in range(3):
    red.zero_grad()
    torch.manual_seed(10 * step + rank)
    model(torch.randn(5, 8)).square().mean().backward()
    red.finish()
g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
both = [torch.zeros_like(g) for _ in range(2)]
torch.distributed.all_gather(both, g)
ws = [torch.zeros_like(w0) for _ in range(2)]
torch.distributed.all_gather(ws, w0)
assert torch.equal(both[0], both[1]) and torch.equal(ws[0], ws[1]) and len(red.buckets) >= 2
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
ddp.COMM.all_reduce(t, "max")
assert float(t) == 2.0
ddp.COMM.barrier()
ddp.shutdown()
print("RANK_OK %%d" %% rank, flush=True)
"""


def test_reducer_under_the_drivers_launcher_gloo_world2(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 (the route the driver takes for bench.py --gpus N): the environment it
    hands the ranks, the control-plane process group over its store, the bucket reducer, the max-reduce bench.py uses for the
    step time, shutdown"""
    import subprocess
    script = tmp_path / "rank.py"
    script.write_text(_TORCHRUN_SCRIPT % {"product": PRODUCT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0 and "RANK_OK 0" in r.stdout and "RANK_OK 1" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_transposed_weight_gradient_plan_round_trips_through_the_plan_file():
    """plan impl 5 (the wide 1x1 layers' weight gradient as a forward GEMM on transposed operands) lives in the host module, not in the
    library's table: export -> reset -> load must bring it back; its applicability rule is what DESIGN 3.8 states"""
    from sqd import nnkernels
    nnkernels.reset_plans()
    try:
        wkey = (5120, 1, 1, 768, 3072, 1, 1)
        nnkernels._register_wgrad_plan(wkey, (nnkernels.WGRAD_TRANSPOSED, 0))
        nnkernels._register_wgrad_plan((12, 48, 160, 64, 64, 3, 3), (1, 28))
        rec = nnkernels.export_plans()
        assert {"pass": "wgrad", "geom": list(wkey), "plan": [5, 0]} in rec["plans"]
        nnkernels.reset_plans()
        assert not nnkernels.CHOSEN_PLANS
        nnkernels.load_plans(rec)
        assert nnkernels.CHOSEN_PLANS[("wgrad",) + wkey] == (5, 0) and nnkernels.CHOSEN_PLANS[("wgrad", 12, 48, 160, 64, 64, 3, 3)] == (1, 28)
        assert nnkernels.plan_mix()["wgrad"] == {"transposed forward-gemm (its own fwd plan)": 1, "fp32 direct": 1}
    finally:
        nnkernels.reset_plans()
    ok = nnkernels.wgrad_transposed_applies
    assert ok((5120, 1, 1, 768, 3072, 1, 1, 1, 0, 1, 1)) and ok((4, 10, 32, 6144, 1536, 1, 1, 1, 0, 10, 32))
    assert not ok((20480, 1, 1, 384, 1536, 1, 1, 1, 0, 1, 1))            # too many rows: the transposes outweigh the product
    assert not ok((12, 12, 40, 256, 1024, 1, 1, 1, 0, 12, 40))           # ResNet-50 layer 3: too narrow
    assert not ok((4, 20, 64, 768, 768, 3, 3, 1, 1, 20, 64))             # not a 1x1 layer
    nnkernels.set_conv_precision(2)                                       # --sqd_bf16: weight gradients stay fp32, the transposed product would not
    try:
        assert not ok((5120, 1, 1, 768, 3072, 1, 1, 1, 0, 1, 1))
    finally:
        nnkernels.set_conv_precision(0)
