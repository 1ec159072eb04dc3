"""bench.py's legs, split by workload (round 6; bench.py is the thin command line over these).

    common         peaks, stdout discipline, scene, rank context, PMC-traffic lookup
    cpu_baselines  the oracle timed on the host cores (cpu_baseline legs), index_parity
    legs_mcl       headline + multinomial + configs[4] + sharded MCL legs
    legs_fastslam  configs[2] / configs[3] + sharded FastSLAM legs
    legs_small     the reference's own sizes, synchronous try_step
    line           emit / compact_line
"""
