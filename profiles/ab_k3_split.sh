# (record of a dropped experiment: the ss_* kernels and FHX_K3_SPLIT are not in the tree; result: profiles/history/r04_y_k3_split_ab.txt)
# K3's large sort as ONE partition into 8 192 buckets by sampled splitters (counts per (bucket, workgroup) in LDS, binary search over
# the splitters in LDS as the digit, 12-byte records written straight to their bucket) + a bitonic LDS sort per bucket, against the
# eight radix passes, on bench.py --overdispersion 1.0 (1.5e7 of 1.2e8 rows below the cutoff)
