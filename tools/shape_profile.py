"""Per-shape cost table of one edit (GPU box):  python tools/shape_profile.py [--num-step 50] [--out gpurun_out/shape_profile.txt]

Records every C-ABI launch of one eagerly executed FreeFine_generation edit (name + shape key + a copy of its arguments), then
replays each UNIQUE launch R times inside a hipGraph and times the graph, so small kernels are priced without host launch
overhead.  Prints calls x per-launch time per shape, sorted by total -- the list the kernel work is planned from."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from freefine_amd import _lib as L  # noqa: E402
from freefine_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--num-step", type=int, default=10)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--model", default="sd21-base")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--out", default="gpurun_out/shape_profile.txt")
ap.add_argument("--no-dedup", action="store_true")
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()

args = argparse.Namespace(model=a.model, vae="sd", dtype=a.dtype, no_graph=True, no_dedup=a.no_dedup, num_step=a.num_step, start_step=0, batch=a.batch)
dev = torch.device("cuda:0")
model = bench.build_model(args, dev, 0, 1)
bench.edit_once(model, args, 0)
torch.cuda.synchronize()
lib = L.load()

REC = {}     # key -> [count, fname, args]


def key_of(fname, args):
    if fname == "ffn_igemm":
        d = args[2]._obj
        return (fname, args[1], "conv" if d.conv else "dense", d.M, d.N, d.K, d.Cin if d.conv else 0, d.stride, d.upsample, d.flags, d.splitk,
                bool(d.residual), bool(d.rowbias))
    if fname == "ffn_attn":
        d = args[2]._obj
        ent = tuple((bool(d.e[i].kmask), bool(d.e[i].qsel), bool(d.e[i].wq), d.e[i].flags, d.e[i].w_const != 0 or d.e[i].w_slope != 0)
                    for i in range(d.npass * L.ATT_MAXB) if (i % L.ATT_MAXB) < d.Bo)
        return (fname, args[1], d.Bo, d.S, d.Sk, d.heads, d.D, d.npass, ent)
    return (fname,) + tuple(x for x in args[1:] if isinstance(x, (int, float)) and not (isinstance(x, int) and x > (1 << 32)))


def copy_args(args):
    out = []
    for x in args:
        if hasattr(x, "_obj"):
            o = x._obj
            out.append(type(o).from_buffer_copy(o))
        else:
            out.append(x)
    return out


def wrap(fname):
    fn = getattr(lib, fname)

    def g(*args):
        k = key_of(fname, args)
        r = REC.get(k)
        if r is None:
            REC[k] = [1, fname, copy_args(args)]
        else:
            r[0] += 1
        return fn(*args)
    return fn, g


LAUNCHERS = [n for n, (_, at) in L.SYMBOLS.items() if at and at[0] is C.c_void_p and n not in ("ffn_device_info",)]
orig = {}
for n in LAUNCHERS:
    orig[n], w = wrap(n)
    setattr(lib, n, w)
# the recorded launches point at transient torch tensors: keep every allocation of the recorded edit alive for the replays
KEEP = []
_empty, _zeros, _empty_like = torch.empty, torch.zeros, torch.empty_like


def _keep(f):
    def g(*x, **kw):
        t = f(*x, **kw)
        KEEP.append(t)
        return t
    return g


torch.empty, torch.zeros, torch.empty_like = _keep(_empty), _keep(_zeros), _keep(_empty_like)
bench.edit_once(model, args, 1)
torch.cuda.synchronize()
torch.empty, torch.zeros, torch.empty_like = _empty, _zeros, _empty_like
for n in LAUNCHERS:
    setattr(lib, n, orig[n])
print(f"recorded {sum(r[0] for r in REC.values())} launches, {len(REC)} unique; kept {len(KEEP)} tensors, "
      f"{torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)

rows = []
for k, (cnt, fname, cargs) in REC.items():
    fn = orig[fname]
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            s = ops._stream()
            for _ in range(a.reps):
                rc = fn(s, *[C.byref(x) if isinstance(x, C.Structure) else x for x in cargs[1:]])
                assert rc == 0, (fname, rc)
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / a.reps)
    except Exception as ex:  # noqa: BLE001
        print("replay failed", k, ex)
        continue
    name = fname
    flops = 0.0
    if fname == "ffn_igemm":
        d = cargs[2]
        buf = C.create_string_buffer(256)
        lib.ffn_igemm_kernel_name(cargs[1], C.byref(d), buf, 256)
        name = buf.value.decode().replace("void igemm_glds_kernel", "ig").replace("(ffn_igemm_desc)", "")
        flops = 2.0 * d.M * d.N * d.K
    elif fname == "ffn_attn":
        d = cargs[2]
        nt = sum(1 for e in k[-1] if e[4])
        flops = 4.0 * nt * d.S * d.Sk * d.heads * d.D
    rows.append((cnt * best, cnt, best, flops, name, k))

rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
with open(a.out, "w") as f:
    hdr = f"# one {a.model} {a.dtype} edit x batch {a.batch}, n={a.num_step}, dedup={not a.no_dedup}: {sum(r[1] for r in rows)} launches, modelled kernel time {tot / 1e3:.1f} ms\n"
    f.write(hdr)
    f.write("total_ms\tshare\tcalls\tus_per_launch\tTFLOP/s\tkernel\tshape_key\n")
    for t, cnt, us, fl, name, k in rows:
        f.write(f"{t / 1e3:9.2f}\t{100 * t / tot:5.1f}%\t{cnt:6d}\t{us:9.2f}\t{fl / us / 1e6 if fl else 0:7.1f}\t{name}\t{k[1:]}\n")
print(open(a.out).read()[:9000])
