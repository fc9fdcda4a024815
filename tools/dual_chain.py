"""Experiment: one DDIM step of B samples as C independent chains of B/C samples on C streams, captured in one hipGraph
(fork/join), against the single-chain graph.

    python tools/dual_chain.py --latent 64 --batch 16 --chains 1 2 4"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from bench import StepRunner, build_model
    dev = torch.device("cuda", 0)
    model, ucfg = build_model(a.latent, dev)
    for c in a.chains:
        per = a.batch // c
        runs = [StepRunner(model, ucfg, per, graph=False, seed=1 + i) for i in range(c)]
        streams = [torch.cuda.Stream() for _ in range(c)]
        for r in runs:
            r.eager_step()
            r.eager_step()
        torch.cuda.synchronize()

        def fn():
            cur = torch.cuda.current_stream()
            for r, s in zip(runs, streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    r.eager_step()
            for s in streams:
                cur.wait_stream(s)

        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        print(f"latent {a.latent} batch {a.batch} as {c} chain(s) of {per}: {ms:.3f} ms/step = {a.batch * 1e3 / ms:.1f} sample-steps/s",
              flush=True)
        del g, runs


if __name__ == "__main__":
    main()
