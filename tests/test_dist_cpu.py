"""CPU, world_size 2 over gloo: the sharding / weight-broadcast / result-gather helpers of the multi-GPU harness."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["FF_ROOT"])
from freefine_amd import dist as FD
from freefine_amd.config import UNetConfig
from freefine_amd.weights import synthetic_state, unet_param_shapes
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
shapes = unet_param_shapes(UNetConfig.preset("tiny"))
ref = synthetic_state(shapes, 0)
got = FD.broadcast_state(ref if rank == 0 else None, shapes, "cpu", chunk_elems=1 << 20)
assert set(got) == set(ref) and all(torch.equal(got[k], ref[k]) and got[k].dtype == torch.float32 for k in ref)
# bf16 payload for the weight matrices (fast mode): every rank receives RNE-rounded matrices, 1-D tensors stay fp32 exactly
got16 = FD.broadcast_state(ref if rank == 0 else None, shapes, "cpu", chunk_elems=1 << 20, matrix_dtype=torch.bfloat16)
for k in ref:
    if ref[k].ndim >= 2:
        assert got16[k].dtype == torch.bfloat16 and torch.equal(got16[k], ref[k].to(torch.bfloat16)), k
    else:
        assert got16[k].dtype == torch.float32 and torch.equal(got16[k], ref[k]), k
assert FD.broadcast_object({"a": rank} if rank == 0 else None) == {"a": 0} and FD.active()
n = 7
mine = FD.shard_indices(n, rank, world)
res = FD.gather_results([{"key": i, "rank": rank, "val": i * i} for i in mine])
assert sorted(r["key"] for r in res) == list(range(n)), res
# the igemm tuning table travels from rank 0 to everyone (identical tile / split-K choices on all ranks)
from freefine_amd import ops
from freefine_amd import _lib
w = ops.tune_table_export().shape[1]
t = torch.zeros(3, w, dtype=torch.int32)
t[:, 0] = _lib.load().ffn_igemm_tune_stamp()
t[:, 1] = torch.tensor([4096, 8192, 98304]) + rank; t[:, 2] = 320; t[:, 3] = 320; t[:, -2] = torch.tensor([1, 6, 13]); t[:, -1] = 1
assert ops.tune_table_import(t) == 3           # every rank has "tuned" its own (different) keys before the sync
n_tab = FD.sync_tune_table()
tab = ops.tune_table_export()
# rank 0's table REPLACES the local one (no rank-local leftovers), and timing-based tuning is frozen on every rank
assert n_tab == 3 and tab.shape[0] == 3 and sorted(tab[:, 1].tolist()) == [4096, 8192, 98304] and sorted(tab[:, -2].tolist()) == [1, 6, 13], tab
assert ops.tune_enable(False) is False          # frozen ...
FD.restore_tuning()                              # ... until restored: back to the setting the freeze replaced (tuning on)
assert ops.tune_enable(True) is True and FD.sync_tune_table.previous is None
with FD.frozen_tuning():
    assert ops.tune_enable(False) is False
assert ops.tune_enable(True) is True
if rank == 0:
    print("DIST_OK", len(res))
dist.destroy_process_group()
'''


def test_shard_indices_matches_distributed_sampler():
    from torch.utils.data import DistributedSampler
    from freefine_amd.dist import shard_indices
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                if n == 0:
                    assert shard_indices(n, rank, world) == []
                    continue
                s = DistributedSampler(list(range(n)), num_replicas=world, rank=rank, shuffle=False, drop_last=False)
                assert list(iter(s)) == shard_indices(n, rank, world), (n, world, rank)


def test_world2_gloo_broadcast_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FF_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIST_OK 7" in out.stdout


def test_bench_gpus_flag_launches_ranks(monkeypatch):
    """`python bench.py --gpus N` (no rank environment) starts N ranks through torch.distributed.run as CHILD processes and relays their exit code;
    a process that already is a rank (WORLD_SIZE set: the driver's own torchrun form, and the children) never re-launches; N = 1 never launches.
    Reference: /root/reference/evaluation/FreeFine/run_script_2D.sh:12-14 (torchrun --nproc_per_node=8)."""
    import argparse
    import importlib.util
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "2", "--warmup", "1"], port=29611)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    calls = []

    def fake_run(c, env=None, **kw):
        calls.append((c, env))
        return subprocess.CompletedProcess(c, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    ns = argparse.Namespace(gpus=4)
    assert bench.maybe_launch_ranks(ns, ["--gpus", "4"], environ={"PATH": "x"}) == 7 and len(calls) == 1
    assert "--nproc-per-node=4" in calls[0][0] and calls[0][1]["PATH"] == "x"
    assert bench.maybe_launch_ranks(ns, ["--gpus", "4"], environ={"WORLD_SIZE": "4", "RANK": "1"}) is None and len(calls) == 1
    assert bench.maybe_launch_ranks(argparse.Namespace(gpus=1), [], environ={}) is None and len(calls) == 1


FOLDER_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["FF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FF_ROOT"], "tools"))
from freefine_amd.pipeline import FreeFinePipeline
from freefine_amd.weights import synthetic_state, unet_param_shapes, vae_param_shapes
dist.init_process_group("gloo")
rank = dist.get_rank()
folder = os.environ["FF_FOLDER"]
# rank 1 must NOT read the tensors: hide them from it (configs / tokenizer / text encoder are small files every rank reads itself)
if rank == 1:
    import freefine_amd.pipeline as P
    real = P.load_safetensors_dir
    def guard(*a, **k):
        raise AssertionError("a non-lead rank read the checkpoint tensors")
    P.load_safetensors_dir = guard
ucfg, ust, vcfg, vst, tok, enc, sched, dtype = FreeFinePipeline.components(folder, torch.float32, "cpu", broadcast="auto")
ref_u = synthetic_state(unet_param_shapes(ucfg), 9)
ref_v = synthetic_state(vae_param_shapes(vcfg), 10)
assert ucfg.heads == (2, 4, 4, 4) and sched.config.steps_offset == 1
assert set(ust) == set(ref_u) and all(torch.equal(ust[k], ref_u[k]) for k in ref_u), "unet bits differ on rank %d" % rank
assert set(vst) == set(ref_v) and all(torch.equal(vst[k].reshape(ref_v[k].shape), ref_v[k]) for k in ref_v), "vae bits differ on rank %d" % rank
# fast mode: the matrices travel as bf16 (what the packers round them to anyway), 1-D parameters as fp32
_, u16, *_ = FreeFinePipeline.components(folder, torch.bfloat16, "cpu", broadcast="auto")
k = "conv_in.weight"
assert u16[k].dtype == torch.bfloat16 and torch.equal(u16[k], ref_u[k].to(torch.bfloat16)) and u16["conv_in.bias"].dtype == torch.float32
if rank == 0:
    print("FOLDER_OK")
dist.destroy_process_group()
'''


def test_world2_gloo_checkpoint_folder_broadcast(tmp_path):
    """from_pretrained(<HF folder>, broadcast="auto") under a 2-rank process group: rank 0 reads the safetensors, rank 1 receives the same bits
    (fp32) / the same bf16 roundings (fast mode) without touching the tensor files; configs and scheduler constants agree on both."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_checkpoint as M
    folder = str(tmp_path / "sd_tiny")
    M.write(folder, "tiny", "tiny", "fp32", seed=9)
    script = tmp_path / "worker_folder.py"
    script.write_text(FOLDER_WORKER)
    env = dict(os.environ, FF_ROOT=ROOT, FF_FOLDER=folder)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29519", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "FOLDER_OK" in out.stdout
