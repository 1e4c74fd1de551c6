"""dev tool: the cluster decode (variant 5) against the stepped decode (variant 3) on fixtures; prints max differences and timings."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from conftest import load_golden
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

def fixture(name, max_steps=None):
    cfg, d = load_golden(name)
    sd = synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"]))
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    soft = cfg.kind == "softmax"
    if soft:
        from oracle import cpu_ref
        n_steps = int(d["n_samples"].max())
        q = d["q"] if "q" in d else cpu_ref.softmax_noise(cfg, n_steps, d["aux"].shape[0], generator=torch.Generator().manual_seed(int(d["noise_seed"])))
        noise = torch.from_numpy(q)
    else:
        noise = torch.from_numpy(d["noise"])
    n_steps = noise.shape[0] if max_steps is None else min(noise.shape[0], max_steps)
    noise = noise[:n_steps].permute(1, 0, 2).contiguous()
    aux = torch.from_numpy(d["aux"])
    a, ha = net.decode(aux, n_steps, noise, want_heads=True, variant=3)
    t0 = time.time()
    b, hb = net.decode(aux, n_steps, noise, want_heads=True, variant=5)
    torch.cuda.synchronize()
    print(name, "steps", n_steps, "cluster %.3f s" % (time.time() - t0), end=" ")
    if soft:
        print("indices equal:", bool((a == b).all().item()), "heads maxdiff %.2e" % (ha - hb).abs().max().item(), "min idx", int(b.min()))
    else:
        print("samples maxdiff %.2e heads maxdiff %.2e nan=%d" % ((a - b).abs().max().item(), (ha - hb).abs().max().item(), int(torch.isnan(b).sum())))

def timing(cfg, B, Tf, variants=(3, 5), reps=3):
    sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    soft = cfg.kind == "softmax"
    seg = 1 if soft else cfg.seg
    outs = {}
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    cond = net.frontend(aux)
    for v in variants:
        res = []
        for n_steps in (Tf * cfg.U // seg // 2, Tf * cfg.U // seg):
            net.decode(aux, n_steps, None, cond=cond, variant=v, rng_seed=5)
            best = 1e9
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.time()
                out, _ = net.decode(aux, n_steps, None, cond=cond, variant=v, rng_seed=5)
                torch.cuda.synchronize(); best = min(best, time.time() - t0)
            res.append((n_steps, best))
        outs[v] = out
        us = (res[1][1] - res[0][1]) / (res[1][0] - res[0][0]) * 1e6
        print(f"{cfg.kind} H={cfg.H} seg={seg} B={B}: variant {v}: {res[1][1]*1e3:.1f} ms for {res[1][0]} steps, steady {us:.1f} us/step", flush=True)
    if len(outs) == 2:
        a, b = outs[variants[0]], outs[variants[1]]
        print("   agree:", float((a == b).float().mean()) if soft else float((a - b).abs().max()))

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    if what == "tiny":
        for n in ("g0_tiny_lap_s1l0_trained", "g0_tiny_lap_s5l4_trained", "g0_tiny_lap_s2l4_xavier", "g0_tiny_softmax", "g0_tiny_softmax_wav"):
            fixture(n)
    elif what == "ref6":
        fixture("g2_ref6_lap_s1l4_b1", 300); fixture("g2_ref6_lap_s5l4_b2", 80); fixture("g2_ref6_softmax_b1", 300)
    elif what == "time":
        timing(C.ref6_laplace(1, 4), 1, 4); timing(C.ref6_laplace(1, 4), 8, 4); timing(C.ref6_laplace(1, 4), 64, 4)
        timing(C.ref6_laplace(5, 4), 1, 4); timing(C.ref6_softmax(), 1, 4)
    elif what == "time1":          # single utterances, long enough for a stable slope
        v = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (3, 5)
        timing(C.ref6_laplace(1, 4), 1, 16, v); timing(C.ref6_laplace(5, 4), 1, 16, v); timing(C.ref6_softmax(), 1, 16, v)
        timing(C.ref6_laplace(1, 4), 2, 16, v)
