"""Parameter grids shared by the test modules — the reference's own
(lol/Crypto/Lol/Tests/Default.hs:45-125) plus BASELINE.json's configurations."""
ZQ1 = [18869761]
ZQ2 = [19393921, 18869761]
ZQ3 = [19918081, 19393921, 18869761]
SMOOTH1 = [2148249601]
SMOOTH3 = [2148854401, 2148249601, 2150668801]

TENSOR1 = [(7, [29]), (12, SMOOTH1), (1, [17]), (2, [17]), (4, [17]), (8, [17]), (21, [8191]),
           (42, [8191]), (42, ZQ1), (2, ZQ2), (3, ZQ2), (7, ZQ2), (6, ZQ2), (42, SMOOTH3),
           (42, ZQ2), (89, [179])]
TENSOR2 = [(1, 7, [29]), (4, 12, [536871001]), (4, 12, SMOOTH1), (2, 8, [17]), (8, 8, [17]),
           (2, 8, SMOOTH1), (4, 8, [17]), (3, 21, [8191]), (7, 21, [8191]), (3, 42, [8191]),
           (3, 21, ZQ1), (7, 21, ZQ2), (3, 42, ZQ3)]
PRIMEOPS_ONLY = [(7, [32]), (42, [1024]), (28, [8]), (91, [4]), (448, [16])]
BIG = [(1024, [12289], 1), (2 ** 14, [1073872897], 2), (15015, [1073842771], 4)]
PRIME_OPS = ("l", "linv", "gpow", "gdec", "ginvpow", "ginvdec")
# oracle method name -> lol_amd.Plan method name (the reference's Tensor class names)
PLAN_NAME = {"crt": "crt", "crtinv": "crtInv", "l": "l", "linv": "lInv", "gpow": "mulGPow",
             "gdec": "mulGDec", "ginvpow": "divGPow", "ginvdec": "divGDec"}
# benchmark parameter sets of the reference (lol/Crypto/Lol/Benchmarks/Default.hs:42-50)
BENCH1 = [(1024, 12289), (2048, 12289), (64 * 27, 3457), (64 * 81, 10369), (64 * 9 * 25, 14401)]
BENCH2 = [(728, 2912, 8737), (728, 3640, 14561), (128, 11648, 23297)]
