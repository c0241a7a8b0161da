"""Root-cause probe for the round-5 finding "the wave-private split-K tiles (71..76) broke the fused-stem equivalence tests under a random
variant mixture".  Run against the probe library (make -C yolopoint_amd/csrc probewsk; YP_HIP_LIB=yolopoint_amd/lib/ab/libPW.so).

One arm = one process (the tuner cache is process-wide): builds the fused-stem plan and the two-launch plan of
tests/test_gpu_model.py::test_fused_stem_conv2_equals_the_two_launches under YP_TUNE_RANDOM=<seed> with the WSK tiles among the candidates, and
prints, per head, max|a - b| / max|b| in units of the test's tolerance, which layers of WHICH plan drew a WSK tile, and the same comparison of
each plan against a default-variant build of itself (summation-order noise shows up as a few 16-bit ulps everywhere; a corrupted buffer as O(1))."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BASE = "1,2,3,4,5,21,22,23,24,25,26,27,10,11,12,13,14,15,16,17,18,19,41,42,43,44,57,58,61,62"
WSK = "71,72,73,74,75,76"


def arm(which, dtype, env_key):
    import torch
    from helpers import make_model
    from oracle import net_oracle
    dev = torch.device("cuda:0")
    outs = {}
    for fuse in ("1", "0"):
        os.environ[env_key] = fuse
        m, _ = make_model("s", 29, dtype=dtype)
        m = m.to(dev).eval()
        m.fuse()
        x = net_oracle.synth_image(2, 3, 160, 224, 31).to(dev)
        with torch.no_grad():
            o = m(x)
        outs[fuse] = [o["semi"].float().cpu(), o["desc"].float().cpu(), o["objects"][0].float().cpu()]
    tol = lambda b: 2.0 ** -7 * float(b.abs().max()) * (1 if dtype == "bf16" else 0.125) + 1e-6
    rec = {"which": which, "ratio_to_tol": [round(float((a - b).abs().max()) / tol(b), 3) for a, b in zip(outs["1"], outs["0"])],
           "frac_differing": [round(float((a != b).float().mean()), 4) for a, b in zip(outs["1"], outs["0"])]}
    torch.save(outs, f"/tmp/wsk_{which}.pt")
    print("RESULT " + json.dumps(rec), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "arm":
        return arm(sys.argv[2], sys.argv[3], sys.argv[4])
    import torch
    lib = os.path.join(ROOT, "yolopoint_amd", "lib", "ab", "libPW.so")
    seeds = [int(s) for s in os.environ.get("SEEDS", "1,2,3,4,5,6,7,8").split(",")]
    for dtype in ("f16",):
        for env_key in ("YP_FUSE_STEM2", "YP_FUSE_STEM3"):
            ref = None
            for seed in [None] + seeds:
                for cands, tag in ((BASE + "," + WSK, "wsk"), (BASE, "base")):
                    if seed is None and tag == "wsk":
                        continue
                    env = dict(os.environ, YP_HIP_LIB=lib, YP_TUNE_ONLY=cands, YP_TUNE_DEBUG="1")
                    if seed is not None:
                        env["YP_TUNE_RANDOM"] = str(seed)
                    which = f"{env_key[-5:]}_{dtype}_{tag}_{seed}"
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "arm", which, dtype, env_key], env=env, capture_output=True, text=True, timeout=900)
                    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
                    picks = [l for l in r.stdout.splitlines() if "tune-random" in l and any(f"pick {t} " in l for t in WSK.split(","))]
                    if not res:
                        print(which, "FAILED", r.stderr[-600:])
                        continue
                    rec = json.loads(res[0][7:])
                    cur = torch.load(f"/tmp/wsk_{which}.pt")
                    if seed is None:
                        ref = cur
                    else:
                        # each plan of this mixture against the default-variant build of the SAME plan
                        tol = lambda b: 2.0 ** -7 * float(b.abs().max()) * 0.125 + 1e-6
                        rec["vs_default_fused"] = [round(float((a - b).abs().max()) / tol(b), 3) for a, b in zip(cur["1"], ref["1"])]
                        rec["vs_default_two_launch"] = [round(float((a - b).abs().max()) / tol(b), 3) for a, b in zip(cur["0"], ref["0"])]
                    rec["wsk_picks"] = [" ".join(p.split()[1:5]) + " " + p.split("k=")[1] if "k=" in p else p for p in picks]
                    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
