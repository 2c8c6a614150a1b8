"""debug: one LI update with the down-date consumers on / off; where does P differ? (64-column block map)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pre3 = importlib.import_module("3pre_amd")
synth = importlib.import_module("3pre_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
H = 200 if N >= 500 else 60
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seq = synth.make_sequence(N, STEPS, H, seed=31 + N)
res = []
for on in (False, True):
    f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=H, std_z=1.0)
    f.k9_overlap(on)
    f.set_x_p_k_k(seq["x0"], seq["P0"])
    for t, s in enumerate(seq["steps"]):
        st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0 if t % 2 == 0 else 0.5, early_exit=False)
    res.append((st, f.get_x_k_k(), f.get_p_k_k()))
    f.close()
(s0, x0, P0), (s1, x1, P1) = res
print(s0); print(s1)
n = P0.shape[0]
D = np.abs(P0 - P1)
print("max |dP|", D.max(), "scale", np.abs(P0).max(), "n", n, "differing entries", int((D > 0).sum()))
nb = (n + 63) // 64
M = np.zeros((nb, nb), int)
for i in range(nb):
    for j in range(nb):
        M[i, j] = int((D[64 * i:64 * i + 64, 64 * j:64 * j + 64] > 0).sum())
np.set_printoptions(linewidth=250, threshold=100000)
print(M)
