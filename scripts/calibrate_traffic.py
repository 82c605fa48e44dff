#!/usr/bin/env python3
"""Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (gfx950) per access shape: kernels that move a
known number of bytes (scripts/calib/calibrate_traffic.hip). Two modes:

  calibrate_traffic.py run                  launches every case 3 times (wrap this in rocprofv3 --pmc FETCH_SIZE
                                            and again in --pmc WRITE_SIZE; counters in separate passes)
  calibrate_traffic.py report <db> [<db>]   reads the rocpd databases of those passes and prints, per case, the bytes
                                            the kernel moved, the counter value and their ratio -- the factor to apply

The cases: streaming reads at 16 / 8 / 4 B per lane, streaming write at 16 B per lane, gathers of 32-byte and 64-byte
records with a stencil-like index (identity + small offsets: consecutive lanes read consecutive records, every record
read 9 times by 9 launches' worth of columns) and with a random permutation (every record read once, no locality)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "ryujin_amd", "lib", "libryujin_calib.so")
SRC = os.path.join(ROOT, "scripts", "calib", "calibrate_traffic.hip")
N = 1 << 26            # elements of the streaming cases (1 GiB at 16 B)
N_REC = 1 << 22        # records of the gather cases (4 M, as the node count of BASELINE configs[1..2])

CASES = [  # (label, which, bytes moved per launch)
    ("stream read 16 B/lane", 0, N * 16), ("stream read 8 B/lane", 1, N * 8), ("stream read 4 B/lane", 2, N * 4),
    ("stream write 16 B/lane", 3, N * 16),
    ("gather 32-B records, stencil order (x9)", 4, None), ("gather 64-B records, stencil order (x9)", 5, None),
    ("gather 32-B records, random order (x1)", 4, None), ("gather 64-B records, random order (x1)", 5, None),
]


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC,
                        "-o", SO], check=True)
    return SO


def run():
    lib = C.CDLL(build())
    lib.ryujin_calib_alloc.restype = C.c_void_p
    lib.ryujin_calib_alloc.argtypes = [C.c_size_t]
    lib.ryujin_calib_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.ryujin_calib_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    buf = lib.ryujin_calib_alloc(N * 16)
    out = lib.ryujin_calib_alloc(N * 16)
    rec = lib.ryujin_calib_alloc(N_REC * 64)
    rng = np.random.default_rng(1)
    nx = 2048
    # stencil order: 9 columns of a 2-D Q1 stencil on a lexicographic numbering, one launch per column
    offsets = [dx + nx * dy for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    base = np.arange(N_REC, dtype=np.int64)
    idx_dev = lib.ryujin_calib_alloc(N_REC * 4)
    for rep in range(3):
        for which, n in ((0, N), (1, N), (2, N), (3, N)):
            assert lib.ryujin_calib_run(which, buf, None, out, n) == 0
        for which in (4, 5):
            for off in offsets:
                idx = np.clip(base + off, 0, N_REC - 1).astype(np.uint32)
                lib.ryujin_calib_upload(idx_dev, idx.ctypes.data, idx.nbytes)
                assert lib.ryujin_calib_run(which, rec, idx_dev, out, N_REC) == 0
        perm = rng.permutation(N_REC).astype(np.uint32)
        lib.ryujin_calib_upload(idx_dev, perm.ctypes.data, perm.nbytes)
        for which in (4, 5):
            assert lib.ryujin_calib_run(which, rec, idx_dev, out, N_REC) == 0
    print("calibration launches done")


def report(dbs):
    import sqlite3
    rows = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection "
                                   "order by dispatch_id"):
            rows.setdefault((k, c), []).append(v)
    print("| access shape | bytes per launch | counter | KiB counted | bytes / (KiB counted * 1024) |")
    print("|---|---|---|---|---|")

    def line(label, kernel, counter, nbytes, pick):
        for (k, c), vals in rows.items():
            if kernel in k and c == counter:
                sel = pick(vals)
                if sel:
                    kib = float(np.mean(sel))
                    print(f"| {label} | {nbytes / 1e6:.1f} MB | {counter} | {kib:.4g} | {nbytes / (kib * 1024):.3f} |")
    line("stream read 16 B/lane", "k_calib_stream_read<HIP_vector_type<double, 2u> >", "FETCH_SIZE", N * 16, lambda v: v)
    line("stream read 8 B/lane", "k_calib_stream_read<double>", "FETCH_SIZE", N * 8, lambda v: v)
    line("stream read 4 B/lane", "k_calib_stream_read<float>", "FETCH_SIZE", N * 4, lambda v: v)
    line("stream write 16 B/lane", "k_calib_stream_write", "WRITE_SIZE", N * 16, lambda v: v)
    # per repetition: 9 stencil launches then 1 random launch of each gather kernel
    for rec_bytes, kernel in ((32, "k_calib_gather<32>"), (64, "k_calib_gather<64>")):
        # compulsory traffic of one launch: every record once (+ the 4-byte index per lane)
        compulsory = N_REC * (rec_bytes + 4)
        line(f"gather {rec_bytes}-B records, stencil order, per launch", kernel, "FETCH_SIZE", compulsory,
             lambda v: [x for q, x in enumerate(v) if q % 10 != 9])
        line(f"gather {rec_bytes}-B records, random order, per launch", kernel, "FETCH_SIZE", compulsory,
             lambda v: [x for q, x in enumerate(v) if q % 10 == 9])


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "report":
        report(sys.argv[2:])
    elif len(sys.argv) >= 2 and sys.argv[1] == "build":
        print(build())
    else:
        run()
