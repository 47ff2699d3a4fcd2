import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import synth, engine
from oracle import c_oracle

def rel(a, b): return float(np.linalg.norm(a-b)/max(np.linalg.norm(b), 1e-300))

for cfg, kw in [(1, {}), ('small_slam', {}), (4, {})]:
    if cfg == 'small_slam':
        sc = synth.make_scenario(8, 30, 6, seed=77)
    else:
        sc = synth.make_config(cfg)
    N = sc['n_poses_max']; K = len(sc['trk_off'])-1; M = len(sc.get('slam_anchor_idxs', []))
    t0 = time.time(); oc = c_oracle.visual_update(sc); t_cpu = time.time()-t0
    eng = engine.Engine(N, M, K)
    eng.stage(sc)
    b = eng.msckf_build(sc['sigma_img'])
    print(cfg, 'inlier equal', np.array_equal(b['inlier'], oc['inlier']), int(b['inlier'].sum()), int(oc['inlier'].sum()),
          'gamma rel max', float(np.max(np.abs(b['gamma']-oc['gamma'])/np.abs(oc['gamma']))))
    if M: print('  slam inl', b['inlier_slam'], oc['inlier_slam'], 'gam rel', float(np.max(np.abs(b['gamma_slam']-oc['gamma_slam'])/np.abs(oc['gamma_slam']))))
    T, z = eng.qr_compress()
    corr = eng.apply_update()
    P = eng.download_P()
    print('  P rel', rel(P, oc['P']), 'corr rel', rel(corr, oc['correction']), 'sym', float(np.abs(P-P.T).max()), 'cpu_s', round(t_cpu,3))
    eng.stage(sc)
    t0=time.time(); r = eng.visual_update_staged(sc['sigma_img']); t1=time.time()-t0
    P2 = eng.download_P()
    print('  staged P rel', rel(P2, oc['P']), 'wall ms', round(1e3*t1,3))
    eng.stage(sc)
    tm = eng.bench_staged(sc['sigma_img'], 2, 5)
    print('  bench', json.dumps(tm))
    eng.close()
