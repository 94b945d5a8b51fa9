"""CPU timing of the REFERENCE sampler itself in the build container -- TEST INFRASTRUCTURE ONLY (BASELINE.md section 3,
item 1: median of warm runs through the oracle/ref_harness.py import recipe, seeded noise tape).  The reference's
Python never travels to the GPU box, so bench.py's `cpu_baseline` there is the oracle port (`kind: "port"`); this
script puts the reference's own number beside it for the same shapes.

    python oracle/time_reference.py [runs]        # prints a table; profiles/README.md quotes it
"""
import contextlib
import io
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh  # noqa: E402
from oracle import trace_oracle as orc  # noqa: E402
from trace_amd import synthetic as syn  # noqa: E402
from trace_amd.config import ModelConfig  # noqa: E402
from trace_amd.weights import random_state_dict  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cfg = ModelConfig()
    aps, N = [16], 20                                   # BASELINE config #1: 1 scene x 16 agents x 20 samples
    M = sum(aps) * N
    batch = syn.make_batch(cfg, aps, seed=1)
    sd = random_state_dict(cfg, 0)
    init, steps = syn.make_noise(cfg, sum(aps), N, seed=3)
    tape = [init] + [steps[i] for i in range(cfg.n_diffusion_steps)]
    guide = syn.target_pos_guidance(batch, aps, weight=30000.0, seed=5)
    rh.load_reference()
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())
    with quiet():
        model = rh.build_reference_model(cfg.to_algo_dict(), sd, batch["map_grid_feat"], batch["map_grid_feat_non_cond"])
    cases = [("unguided, w = 0 (BASELINE config #1)", None, 0.0, False),
             ("target_pos, CFG w = 0.5, clean guidance (config #2's arithmetic at M = 320)", guide, 0.5, True)]
    print("threads %d, M = %d trajectories, 100 DDPM steps; median of %d warm runs (1 warm-up)" % (
        torch.get_num_threads(), M, runs))
    for name, g, w, clean in cases:
        with quiet():
            if g is not None:
                model.set_guidance(g, batch)
                model.update_guidance(global_t=0)
            else:
                model.clear_guidance() if hasattr(model, "clear_guidance") else None
        ts = []
        for r in range(runs + 1):
            t0 = time.perf_counter()
            with rh.noise_tape(tape), quiet():
                model(batch, num_samp=N, return_diffusion=False, return_guidance_losses=g is not None,
                      class_free_guide_w=w, apply_guidance=True, guide_clean=clean)
            ts.append(time.perf_counter() - t0)
        ref = statistics.median(ts[1:])
        # the oracle port on the same inputs (what bench.py's cpu_baseline times on the GPU box)
        o = orc.OracleSampler(cfg, sd)
        o.set_guidance(g) if g is not None else o.clear_guidance()
        tp = []
        for r in range(3):
            t0 = time.perf_counter()
            o.sample(batch, N, w, True, clean, return_diffusion=False)
            tp.append(time.perf_counter() - t0)
        port = statistics.median(tp[1:])
        print("%-80s reference %7.2f s = %6.1f trajectories/s   oracle port %7.2f s = %6.1f trajectories/s" % (
            name, ref, M / ref, port, M / port))


if __name__ == "__main__":
    main()
