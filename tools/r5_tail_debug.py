"""Round-5 debug: one step at N=500 through pre3_step with the tail inside k_cholp; the rescue gate's outcome against numpy variants."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pre3 = importlib.import_module("3pre_amd"); synth = importlib.import_module("3pre_amd.synth")
import oracle as orc
from oracle import np_twin as tw

N, n_hyp = int(os.environ.get("DBG_N", 500)), 200
seq = synth.make_sequence(N, 3, n_hyp)
types, off, n = orc.landmark_table(np.zeros(N, int))
f = pre3.EkfFilter(seq["cam"], np.zeros(N, np.int32), dtype="f32", max_hyp=n_hyp, std_z=1.0)
x, P = seq["x0"], seq["P0"]
s = seq["steps"][0]
ref = tw.step(types, off, seq["cam"], x, P, s["u"], s["meas_idx"], s["z"], s["hyp"], 1.0, early_exit=False)
f.set_x_p_k_k(x, P)
st = f.step(s["u"], s["meas_idx"], s["z"], s["hyp"], threshold=1.0, early_exit=False)
li, hi = f.get_flags()
print("stats", st, "ref n_li", int(ref["li"].sum()), "ref n_hi", int(ref["hi"].sum()), "gpu n_hi", int(hi.sum()))
print("li equal", np.array_equal(li, ref["li"]))
fld = f.get_landmark_fields() if hasattr(f, "get_landmark_fields") else None
xg, Pg = f.get_x_k_k(), f.get_p_k_k()
print("x err", np.abs(xg - ref["x_kk"]).max() if "x_kk" in ref else None)
print(sorted(ref.keys()))
# ---- the twin's intermediates
cam = seq["cam"]
meas_idx = np.asarray(s["meas_idx"], np.int64)
x1, P1 = tw.predict(x, P, s["u"])
h, has_h = tw.project(types, off, x1, cam)
Hc, Hl = tw.jacobian(types, off, x1, cam, h, has_h)
z = np.zeros((N, 2)); z[meas_idx] = s["z"]
ic = np.zeros(N, np.int32); ic[meas_idx] = 1
liL = np.zeros(N, np.int32); liL[meas_idx] = ref["li"]
x2, P2 = tw.update_landmarks(types, off, np.nonzero(liL)[0], x1, P1, Hc, Hl, z, h)
h2, has2 = tw.project(types, off, x2, cam, h, has_h)
Hc2, Hl2 = tw.jacobian(types, off, x2, cam, h2, has2)
hi_ref = tw.rescue(types, off, P2, Hc2, Hl2, h2, z, ic, liL)
fld = f.landmark_fields()
print("proj: h err", np.abs(fld["h"] - h2).max(), "Hc err", np.abs(fld["Hc"].reshape(N, 14) - np.asarray(Hc2).reshape(N, 14)).max(), "Hl err", np.abs(fld["Hl"].reshape(N, 12) - np.asarray(Hl2).reshape(N, 12)).max())
hiL = np.zeros(N, np.int32); hiL[meas_idx] = hi
cand = np.nonzero((ic == 1) & (liL == 0))[0]
print("candidates", len(cand), "ref hi", int(hi_ref[cand].sum()), "gpu hi", int(hiL[cand].sum()), "disagree", int((hi_ref[cand] != hiL[cand]).sum()))
# d2 of the reference per candidate, and of variants: gate on P1 (no down-date), gate without the Jnorm pass
def d2(Pm, i):
    c = tw._sparse_cols(types, off, [i]); Hi = tw._rows(n, types, off, [i], Hc2, Hl2)[:, c]
    Si = Hi @ Pm[np.ix_(c, c)] @ Hi.T; nu = z[i] - h2[i]
    return float(nu @ np.linalg.inv(Si) @ nu)
bad = [i for i in cand if hi_ref[i] != hiL[i]][:12]
for i in bad:
    print(i, "ref d2 %.4g" % d2(P2, i), "d2 on P1 %.4g" % d2(P1, i), "gpu", hiL[i], "ref", hi_ref[i])
if os.environ.get("PRE3_LIB", "").endswith("tdbg.so"):
    D = fld["S"].reshape(N, 4)          # debug build: q00, s00, q11, s11 of the gate
    for i in bad[:8] + [int(c) for c in cand[:4]]:
        cc = tw._sparse_cols(types, off, [i]); Hi = tw._rows(n, types, off, [i], Hc2, Hl2)[:, cc]
        Jn = tw.normjac(x2u[3:7]) if False else None
        q_ref = Hi @ P1[np.ix_(cc, cc)] @ Hi.T
        S_ref = Hi @ P2[np.ix_(cc, cc)] @ Hi.T
        print(i, "gpu q00 %.5g s00 %.5g q11 %.5g s11 %.5g | ref (no Jn) q00 %.5g q11 %.5g ; S00 %.5g S11 %.5g" % (D[i, 0], D[i, 1], D[i, 2], D[i, 3], q_ref[0, 0], q_ref[1, 1], S_ref[0, 0], S_ref[1, 1]))
    # host: q'' = (H J) P1 (H J)', s'' = |(H J) W'|^2 with W = L^-1 H_li P1
    liidx = np.nonzero(liL)[0]
    Hli = tw._rows(n, types, off, list(liidx), Hc, Hl)
    HP = Hli @ P1
    S_li = HP @ Hli.T + np.eye(Hli.shape[0])
    Lc = np.linalg.cholesky(S_li)
    W = np.linalg.solve(Lc, HP)
    # x after the LI update, un-normalised quaternion -> normalisation Jacobian
    nu_li = (z[liidx] - h[liidx]).reshape(-1)
    x2u = x1 + W.T @ np.linalg.solve(Lc, nu_li)
    q = x2u[3:7]; r_, x_, y_, z_ = q
    sN = (q @ q) ** -1.5
    Jn = sN * np.array([[x_*x_+y_*y_+z_*z_, -r_*x_, -r_*y_, -r_*z_], [-x_*r_, r_*r_+y_*y_+z_*z_, -x_*y_, -x_*z_], [-y_*r_, -y_*x_, r_*r_+x_*x_+z_*z_, -y_*z_], [-z_*r_, -z_*x_, -z_*y_, r_*r_+x_*x_+y_*y_]])
    Jf = np.eye(n); Jf[3:7, 3:7] = Jn
    for i in bad[:4] + [int(c) for c in cand[:3]]:
        Hi = tw._rows(n, types, off, [i], Hc2, Hl2) @ Jf
        qq = Hi @ P1 @ Hi.T; yy = Hi @ W.T; ss = yy @ yy.T
        print(i, "host q00 %.5g s00 %.5g q11 %.5g s11 %.5g  (S = %.5g %.5g)" % (qq[0, 0], ss[0, 0], qq[1, 1], ss[1, 1], qq[0, 0] - ss[0, 0], qq[1, 1] - ss[1, 1]))
