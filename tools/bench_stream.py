"""Achieved HBM bandwidth of the three stream kernels (replay sequence gather, straight-through categorical
sampling, lambda-return scan) at sizes that fill the device, beside the sizes of the default configuration.

    python tools/bench_stream.py [out.json] [--large-only]

--large-only runs one device-filling case per kernel (the cases a kernel trace or a counter pass of this
command is read against: every launch of a kernel then has the same size).

Bytes are the algorithmic ones (every input read once, every output written once); time is the mean of
back-to-back launches between two events on the launch stream.  Peak: 8 TB/s.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamer_amd import _native as N  # noqa: E402

PEAK_GBS = 8000.0


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def run(large_only=False, device="cuda:0", verbose=True):
    """Times the three kernels and returns one record per case."""
    LARGE = large_only
    dev = torch.device(device)
    L = N.lib()
    S = lambda: N.stream_ptr(dev)
    rows_out = []

    def report(kernel, case, nbytes, sec):
        gbs = nbytes / sec * 1e-9
        rows_out.append({"kernel": kernel, "case": case, "algorithmic_bytes": nbytes, "us": sec * 1e6, "GB/s": gbs,
                         "frac_of_hbm_peak": gbs / PEAK_GBS})
        if verbose:
            print(f"{kernel:18s} {case:44s} {nbytes / 1e6:10.2f} MB {sec * 1e6:9.1f} us {gbs:8.0f} GB/s  {gbs / PEAK_GBS:5.2f}",
                  flush=True)

    # ---------------------------------------------------------------- replay sequence gather
    cap, frame = 100_000, 3 * 64 * 64
    g = torch.Generator(device=dev).manual_seed(1234)
    obs = torch.randint(0, 256, (cap, 3, 64, 64), dtype=torch.uint8, device=dev, generator=g)
    act = torch.randint(0, 6, (cap,), dtype=torch.int64, device=dev, generator=g)
    rew = torch.randn(cap, device=dev, generator=g)
    con = torch.rand(cap, device=dev, generator=g) > 0.01
    ring = N.drm_ring(obs.data_ptr(), act.data_ptr(), rew.data_ptr(), con.data_ptr(), cap, frame)
    rs = np.random.RandomState(42)
    for B, T, f32 in (((1024, 64, False),) if LARGE else
                      ((16, 64, False), (16, 64, True), (256, 64, False), (256, 64, True), (1024, 64, False))):
        starts = torch.from_numpy(rs.randint(0, cap, size=B)).to(dev)
        o = torch.empty((B, T, 3, 64, 64), dtype=torch.uint8, device=dev)
        a = torch.empty((B, T), dtype=torch.int64, device=dev)
        r = torch.empty((B, T), dtype=torch.float32, device=dev)
        c = torch.empty((B, T), dtype=torch.bool, device=dev)
        f = torch.empty((B, T, 3, 64, 64), dtype=torch.float32, device=dev) if f32 else None

        def fn():
            N.check(L.drm_seq_gather(C.byref(ring), starts.data_ptr(), B, T, o.data_ptr(), a.data_ptr(), r.data_ptr(),
                                     c.data_ptr(), N.ptr(f), S()), "drm_seq_gather")
        nbytes = B * T * (2 * (frame + 8 + 4 + 1) + (4 * frame if f32 else 0))
        report("k_seq_gather", f"B={B} T={T}" + (" + f32/255 image" if f32 else ""), nbytes, timed(fn, 50 if B <= 256 else 20))
        del o, f
    del obs

    # ---------------------------------------------------------------- categorical sampling (K = 32)
    for rows, wz, wi, wy in (((4 * 61440 * 32, True, True, False),) if LARGE else
                            ((16 * 32, True, True, False), (61440 * 32, True, True, False), (61440 * 32, False, True, False),
                             (61440 * 32, True, True, True), (4 * 61440 * 32, True, True, False))):
        lg = torch.randn(rows, 32, device=dev, generator=g)
        ex = torch.empty(rows, 32, device=dev).exponential_(generator=g)
        z = torch.empty_like(lg) if wz else None
        ix = torch.empty(rows, dtype=torch.uint8, device=dev) if wi else None
        y = torch.empty_like(lg) if wy else None

        def fn():
            N.check(L.drm_sample_categorical_st(lg.data_ptr(), ex.data_ptr(), rows, 32, 0.01, N.ptr(z), N.ptr(ix), N.ptr(y), S()),
                    "drm_sample_categorical_st")
        nbytes = rows * (256 + (128 if wz else 0) + (128 if wy else 0) + (1 if wi else 0))
        outs = "+".join(n for n, w in (("z", wz), ("idx", wi), ("y", wy)) if w)
        report("k_sample_st", f"rows={rows} ({rows // 32} states x 32 latents) -> {outs}", nbytes, timed(fn, 50))
        del lg, ex, z, y

    # ---------------------------------------------------------------- lambda-return scan
    for B, H in (((1 << 22, 15),) if LARGE else ((16, 15), (4096, 15), (1 << 20, 15), (1 << 22, 15), (1 << 20, 64), (1 << 16, 333))):
        R = torch.randn(B, H, device=dev, generator=g)
        Cc = torch.rand(B, H, device=dev, generator=g)
        V = torch.randn(B, H, device=dev, generator=g)
        VL = torch.randn(B, device=dev, generator=g)
        out = torch.empty_like(R)

        def fn():
            N.check(L.drm_lambda_return(R.data_ptr(), Cc.data_ptr(), V.data_ptr(), VL.data_ptr(), out.data_ptr(), B, H, 0.997, 0.95,
                                        S()), "drm_lambda_return")
        report("k_lambda_return", f"B={B} H={H}", 16 * B * H + 4 * B, timed(fn, 50))
        del R, Cc, V, out

    return rows_out


if __name__ == "__main__":
    ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = run("--large-only" in sys.argv)
    if ARGS:
        with open(ARGS[0], "w") as fh:
            json.dump({"peak_GB/s": PEAK_GBS, "rows": rows}, fh, indent=1)
