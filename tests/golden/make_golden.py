#!/usr/bin/env python3
"""Generates the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the development container only (needs /root/reference and `bash oracle/build_ref.sh`):

    python3 tests/golden/make_golden.py

Every expected value is computed by the reference: its Fortran (fortran/waterlib.f90, compiled
with flang into an f2py module by oracle/build_ref.sh) driven by its own
structureLibs/water_properties.py imported from where it lies (oracle/ref_loader.py).  Inputs are
seeded synthetic boxes (waterorderlib_amd/synth.py).  The fixtures are data only: inputs and
the reference's outputs.

Per case `<name>.npz` holds
  pos[N,3], sub[Ns,3] (or absent when sub is pos), box[3], low, high
  q[Ns]                         wp.getOrderParamq(sub, pos, box, low, high)        (wp:344-391)
  nn_idx[Ns,4], nn_dist[Ns,4], ang[Ns,6]
                                the 4-NN selection and angle set *implied* by that call, obtained
                                by re-running wp:363-386 line by line with the reference's own
                                wl.* routines and numpy (and asserted to reproduce q bit for bit)
  mask_sum                      checksum of wl.(all)nearneighbors
  cos_ang, cos_num              wp.getCosAngs(sub, pos, box)                        (wp:210-250)
  lsi, lsi_num                  wp.getLSI(sub, pos, box)                            (wp:252-311)
  qhist[500], q_mean, q_var     np.histogram / np.mean / np.var as orderParam_lib.py:1473,1493
  ang_hist[500]                 np.histogram(cos_ang, 500, [0,180]) as wp:328
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_loader  # noqa: E402
from waterorderlib_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def implied_selection(wl, sub, pos, box, low, high):
    """wp.getOrderParamq's loop body (water_properties.py:363-388), keeping the intermediates."""
    ns = len(sub)
    if np.array_equal(sub, pos):
        nb = wl.allnearneighbors(pos, box, low, high).astype(bool)
    else:
        nb = wl.nearneighbors(sub, pos, box, low, high).astype(bool)
    q = np.zeros(ns)
    idx = np.full((ns, 4), -1, dtype=np.int32)
    dist = np.full((ns, 4), np.nan)
    ang = np.full((ns, 6), np.nan)
    for i, apos in enumerate(sub):
        cnt = int(np.sum(nb[i]))
        if cnt > 0:
            members = np.where(nb[i])[0]
            this_pos = wl.reimage(pos[nb[i]], apos, box)
            this_d = np.linalg.norm(this_pos - apos, axis=1)
            order = np.argsort(this_d)
            # exact ties would make the reference's order arbitrary: fixtures must not contain any
            sd = this_d[order][:5]
            assert np.all(np.diff(sd) > 0), "exact distance tie in fixture input"
            new_pos = this_pos[order][:4]
            temp = wl.tetracosang(apos, new_pos, box)
            av = temp[np.triu_indices(len(temp), k=1)]
            if cnt == 1:
                av = 180.0 * np.ones(6)
            elif cnt == 2:
                av = np.concatenate((av, 180.0 * np.ones(5)))
            elif cnt == 3:
                av = np.concatenate((av, 180.0 * np.ones(3)))
            q[i] = 1.0 - (3.0 / 8.0) * np.sum((np.cos(av * np.pi / 180.0) + (1.0 / 3.0)) ** 2)
            k = min(cnt, 4)
            idx[i, :k] = members[order][:k]
            dist[i, :k] = this_d[order][:k]
            ang[i] = av
    return q, idx, dist, ang, int(nb.sum())


def make_case(name, wp, wl, pos, box, sub=None, low=0.0, high=10.0, extras=True):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    box = np.asarray(box, dtype=np.float64)
    s = pos if sub is None else np.ascontiguousarray(sub, dtype=np.float64)
    q_ref = wp.getOrderParamq(s, pos, box, low, high)
    q2, idx, dist, ang, msum = implied_selection(wl, s, pos, box, low, high)
    assert np.array_equal(q_ref, q2), name
    d = dict(pos=pos, box=box, low=low, high=high, q=q_ref, nn_idx=idx, nn_dist=dist, ang=ang,
             mask_sum=msum)
    if sub is not None:
        d["sub"] = s
    qh, _ = np.histogram(q_ref, bins=500, range=[0.0, 1.0], density=False)
    d.update(qhist=qh.astype(np.int64), q_mean=np.mean(q_ref), q_var=np.var(q_ref))
    if extras:
        ca, cn = wp.getCosAngs(s, pos, box)
        li, ln = wp.getLSI(s, pos, box)
        ah, _ = np.histogram(ca, bins=500, range=[0.0, 180.0], density=False)
        d.update(cos_ang=ca, cos_num=cn, lsi=li, lsi_num=ln, ang_hist=ah.astype(np.int64))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("%-18s N=%5d Ns=%5d  <q>=%+.4f  neighbours<4: %d" %
          (name, len(pos), len(s), q_ref.mean(), int((idx[:, 3] < 0).sum())))


def main():
    wp, wl = ref_loader.load_reference()

    # (1) 512-water ideal gas, cubic (BASELINE config 1 size)
    pos, box = synth.ideal_gas(512, seed=1)
    make_case("gas512", wp, wl, pos, box)

    # (2) 512-water jittered diamond lattice (water-like q), TIP4P-style frame -> O rows
    xyz, boxes = synth.water_frames(512, 1, seed=2, len_wat=4)
    make_case("ice512", wp, wl, xyz[0, 0::4], boxes[0])
    np.savez_compressed(os.path.join(OUT, "ice512_frame.npz"), xyz=xyz[0], box=boxes[0], len_wat=4)

    # (3) 2000 waters, orthorhombic box, molecules unwrapped (coordinates outside [0,L))
    box3 = np.array([36.0, 41.5, 40.012])
    n3 = 2000
    pos3, _ = synth.water_frames(n3, 1, seed=3, box=box3)
    rng = np.random.default_rng(33)
    shift = rng.integers(-2, 3, size=(n3, 3)).astype(np.float64) * box3
    make_case("ortho2000_unwrapped", wp, wl, pos3[0] + shift, box3)

    # (4) sparse / vacuum: waters with 0,1,2,3 and >=4 neighbours inside 10 A
    L = 80.0
    rng = np.random.default_rng(4)
    cl = []
    centres = rng.random((12, 3)) * (L - 30.0) + 15.0
    centres = np.array([[10 + 20 * (i % 3), 10 + 20 * ((i // 3) % 3), 10 + 25 * (i // 9)] for i in range(12)], float)
    for k, c in enumerate(centres):
        size = [1, 2, 3, 4, 5, 6, 1, 2, 3, 4, 8, 5][k]
        cl.append(c + rng.normal(0, 1.6, size=(size, 3)))
    make_case("sparse_clusters", wp, wl, np.concatenate(cl), np.array([L, L, L + 7.0]))

    # (5) sub-population branch (wl.nearneighbors): subset rows, and points that are not waters
    pos5, box5 = synth.water_frames(1000, 1, seed=5)
    make_case("subpop_rows", wp, wl, pos5[0], box5[0], sub=pos5[0][:100].copy())
    rng = np.random.default_rng(55)
    make_case("subpop_points", wp, wl, pos5[0], box5[0], sub=rng.random((64, 3)) * box5[0] * 1.5 - 5.0)

    # (6) tiny box: cutoff 10 A > L/2, every pair is evaluated through anint(d/L)
    pos6, box6 = synth.ideal_gas(40, seed=6, box=np.array([13.0, 11.0, 17.5]))
    make_case("tiny_box", wp, wl, pos6, box6)

    # (7) slab with a non-periodic axis (BoxL < 0 => iBoxL = 0, waterlib.f90:41)
    pos7, box7 = synth.water_frames(600, 1, seed=7)
    b7 = box7[0].copy(); b7[2] = -1.0
    make_case("nonperiodic_z", wp, wl, pos7[0], b7)

    # (8) lowCut > 0 and a short highCut (fewer than 4 in shell for many waters)
    make_case("cut_2p9_4p0", wp, wl, pos5[0], box5[0], low=2.9, high=4.0, extras=False)

    # (9) two-frame mini trajectory for the tetOrderCalc reduction (orderParam_lib.py:1458-1495)
    xyz9, boxes9 = synth.water_frames(700, 3, seed=9)
    qs = [wp.getOrderParamq(xyz9[t], xyz9[t], boxes9[t]) for t in range(3)]
    qall = np.concatenate(qs)
    qh, _ = np.histogram(qall, bins=500, range=[0.0, 1.0], density=False)
    np.savez_compressed(os.path.join(OUT, "traj700x3.npz"), frames=xyz9, boxes=boxes9,
                        q=np.stack(qs), mean=np.array([np.mean(q) for q in qs]),
                        var=np.array([np.var(q) for q in qs]), qhist=qh.astype(np.int64))
    print("traj700x3          <q> per frame:", [float(np.mean(q)) for q in qs])

    # (10) waterlib-level known answers
    tet = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], float) * (2.75 / np.sqrt(3.0))
    centre = np.array([20.0, 20.0, 20.0])
    posT = np.vstack([centre, centre + tet + np.array([[1e-3, 0, 0], [0, 2e-3, 0], [0, 0, 3e-3], [4e-3, 0, 0]])])
    boxT = np.array([40.0, 40.0, 40.0])
    angT = wl.tetracosang(centre, posT[1:], boxT)
    re = wl.reimage(np.array([[5.0, -5.0, 4.999], [15.0, 25.0, -14.0]]), np.zeros(3), np.array([10.0, 10.0, 10.0]))
    ld = wl.lsidists(centre, posT[1:] + 40.0, boxT)
    np.savez_compressed(os.path.join(OUT, "waterlib_kat.npz"), tet_pos=posT, tet_box=boxT,
                        tet_ang=angT[np.triu_indices(4, k=1)], tet_q=wp.getOrderParamq(posT[:1], posT, boxT),
                        reimage_in=np.array([[5.0, -5.0, 4.999], [15.0, 25.0, -14.0]]), reimage_out=re,
                        lsidists=ld)
    print("waterlib_kat       tet q =", wp.getOrderParamq(posT[:1], posT, boxT))

    # (11) SURVEY 8f row f-1: Willard-Chandler density field of a slab (waterlib.f90:1286-1398)
    xyzW, boxesW = synth.slab_frames(700, 1, seed=12, n_vapour=5)
    posW, boxW = xyzW[0], boxesW[0]
    gx = np.linspace(1.0, boxW[0] - 1.0, 9); gy = np.linspace(0.5, boxW[1], 7); gz = np.linspace(-3.0, boxW[2] + 2.0, 14)
    dv, dn = wl.willarddensityfield(posW, gx, gy, gz, boxW, 2.4)
    pts = np.random.default_rng(3).random((200, 3)) * boxW * 1.2 - 3.0
    pv, pn = wl.willarddensitypoints(posW, pts, boxW, 2.4)
    np.savez_compressed(os.path.join(OUT, "willard_slab700.npz"), pos=posW, box=boxW, gridx=gx, gridy=gy, gridz=gz,
                        smoothlen=2.4, densvals=dv, densnorms=dn, pts=pts, pts_densvals=pv, pts_densnorms=pn)
    print("willard_slab700    max density %.4f, grid points without water in range: %d" % (dv.max(), int(np.isnan(dn[..., 0]).sum())))

    # (12) SURVEY 8f row f-3: radial distribution functions (waterlib.f90:193-231, 316-389) as rdfCalc calls them
    xyzR, boxesR = synth.water_frames(700, 1, seed=31)
    posR, boxR = xyzR[0], boxesR[0]
    solR = np.random.default_rng(5).random((25, 3)) * boxR
    np.savez_compressed(os.path.join(OUT, "rdf700.npz"), pos=posR, box=boxR, sol=solR, binwidth=0.1, totbins=150,
                        rdf_same=wl.radialdistsame(posR, 0.1, 150, 1.0, boxR),
                        rdf_sol_wat=wl.radialdist(solR, posR, 0.1, 150, 0.0334, boxR),
                        rdf_sol_sol=wl.radialdistsame(solR, 0.1, 150, 1.0, boxR),
                        hist_sol_wat=wl.pairdistancehistogram(solR, posR, 0.25, 60, boxR),
                        hist_wat_wat=wl.pairdistancehistogram(posR, posR, 0.05, 300, boxR),
                        rdf_same_nonperiodic_z=wl.radialdistsame(posR, 0.1, 150, 1.0, boxR * np.array([1, 1, -1.0])))
    print("rdf700             first-shell peak g = %.3f (bulkdens 1 -> local density)" % wl.radialdistsame(posR, 0.1, 150, 1.0, boxR)[:40].max())

    # (13) SURVEY 8f row f-2: water orientation (waterlib.f90:954-1044); hydrogens wrapped on their own so the
    #      O-H vectors need the minimum-image step
    xyzO, boxesO = synth.water_frames(500, 1, seed=41, len_wat=3)
    fO, boxO = xyzO[0], boxesO[0]
    Oo = fO[0::3].copy(); Ho = np.empty((1000, 3)); Ho[0::2] = fO[1::3]; Ho[1::2] = fO[2::3]
    Ho = Ho - boxO * np.floor(Ho / boxO)
    refO = np.array([0.3, -0.2, 1.0])
    adO, apO = wl.watorient(Oo, Ho, refO, boxO)
    np.savez_compressed(os.path.join(OUT, "orient500.npz"), opos=Oo, hpos=Ho, box=boxO, refvec=refO, angdip=adO,
                        angplane=apO, normdip=wl.watohvec(Oo, Ho, boxO), angdip_z=wp.waterOrientation(Oo, Ho, boxO)[0])
    print("orient500          <dipole angle> = %.2f deg" % adO.mean())

    # (14) SURVEY 8f row f-4: hydrogen bonds (waterlib.f90:1156-1210), every water oxygen as acceptor, every O-H as donor
    xyzB, boxesB = synth.water_frames(600, 1, seed=51, len_wat=3)
    fB, boxB = xyzB[0], boxesB[0]
    Ob = fB[0::3].copy(); donB = np.repeat(Ob, 2, axis=0); Hb = np.empty((1200, 3)); Hb[0::2] = fB[1::3]; Hb[1::2] = fB[2::3]
    b1 = wl.generalhbonds(Ob, donB, Hb, boxB, 3.0, 150.0); b2 = wl.generalhbonds(Ob, donB, Hb, boxB, 3.5, 120.0)
    accI = np.arange(0, 1800, 3); donI = np.repeat(accI, 2); donHI = (donI.reshape(-1, 2) + np.array([1, 2])).reshape(-1)
    nhb, hbl, hbloc = wp.HBondsGeneral(Ob, donB, Hb, boxB, accI, donI, donHI, distCut=3.5, angCut=120.0)
    np.savez_compressed(os.path.join(OUT, "hbonds600.npz"), acc=Ob, don=donB, donh=Hb, box=boxB,
                        pairs_3p0_150=np.stack(np.where(b1 == 1), axis=1).astype(np.int32),
                        pairs_3p5_120=np.stack(np.where(b2 == 1), axis=1).astype(np.int32),
                        acc_inds=accI, don_inds=donI, donh_inds=donHI, hbg_num=nhb, hbg_list=hbl, hbg_loc=hbloc)
    print("hbonds600          bonds at (3.0 A, 150 deg): %d, at (3.5 A, 120 deg): %d" % (b1.sum(), b2.sum()))


if __name__ == "__main__":
    main()
