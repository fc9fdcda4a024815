"""Diagnostic: what each family of launches really costs INSIDE the captured step (not its rocprof duration): the step's
launch program is captured with one family left out and the replay time is compared with the full step's.  Results of the
reduced programs are garbage (their inputs are missing), timing is what is read.
    python tools/marginal_cost.py --latent 32 --batch 1
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def replay_us(fn, reps=30):
    from dsml_thesis_amd.engine import GraphedProgram
    g = GraphedProgram(fn)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    from bench import StepRunner, build_model
    from dsml_thesis_amd import lib as L
    model, ucfg = build_model(a.latent, torch.device("cuda", 0))
    run = StepRunner(model, ucfg, a.batch, graph=False)
    pg = run.pg
    calls = list(pg.calls)
    names = sorted({c[3] for c in calls})

    def runner(skip):
        def fn():
            st = torch.cuda.current_stream().cuda_stream
            for f, args, _, name in calls:
                if name in skip:
                    continue
                L.check(f(*args, st), name)
        return fn

    full = replay_us(runner(()))
    print(f"full step: {full:8.1f} us, {len(calls)} calls")
    print(f"program as it runs (side launches on their forked branch), UNet only: {replay_us(pg.run):8.1f} us; "
          f"whole DDIM step: {replay_us(run.eager_step):8.1f} us")
    for name in names:
        n = sum(1 for c in calls if c[3] == name)
        if name == "ldmk_igemm":
            continue
        t = replay_us(runner((name,)))
        print(f"without {name:28s} ({n:3d} calls): {t:8.1f} us -> {(full - t) / n:6.2f} us per call")
    # the GEMMs by themselves (everything else left out), and split by whether they carry a reduce launch
    others = tuple(n for n in names if n != "ldmk_igemm")
    t = replay_us(runner(others))
    ng = sum(1 for c in calls if c[3] == "ldmk_igemm")
    print(f"only ldmk_igemm ({ng} calls): {t:8.1f} us -> {t / ng:6.2f} us per call")


if __name__ == "__main__":
    main()
