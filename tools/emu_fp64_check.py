import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import sys, ctypes; sys.path.insert(0,'tests'); sys.path.insert(0,'tests/emu')
import numpy as np
from gymnasium_robotics_amd.envs.adroit_spec import load_adroit_model, make_adroit_task, action_scaling
lib=sys.argv[1]
for task in sys.argv[2:]:
    m=load_adroit_model(task)
    g=np.load(f'tests/golden/adroit_{task}_teacher.npz')
    L=ctypes.CDLL(lib)
    L.emu_create.restype=ctypes.c_void_p; L.emu_create.argtypes=[ctypes.c_void_p]*3
    H,I,F=m.pack(); h=L.emu_create(H.ctypes.data,I.ctypes.data,F.ctypes.data)
    task_s=make_adroit_task(m,"dense",task)
    class T64(ctypes.Structure):
        _fields_=[("n_substeps",ctypes.c_int),("sparse_reward",ctypes.c_int),("kind",ctypes.c_int),("site",ctypes.c_int*5),("obj_body",ctypes.c_int),("nq_obs",ctypes.c_int),("obs_dim",ctypes.c_int),("qadr",ctypes.c_int*2),("len",ctypes.c_double*2)]
    t=T64()
    for f,_ in T64._fields_:
        v=getattr(task_s,f)
        if f in("site","qadr","len"):
            for k in range(len(v)): getattr(t,f)[k]=v[k]
        else: setattr(t,f,v)
    am,ar=action_scaling(m)
    p=lambda a: a.ctypes.data_as(ctypes.c_void_p)
    E=[]
    for i in range(0,420,3):
        qp,qv,qa=g["qpos"][i].astype(np.float64).copy(), g["qvel"][i].astype(np.float64).copy(), g["qacc_ws"][i].astype(np.float64).copy()
        sh=g["shift"][i].astype(np.float64).copy(); tg=g["target"][i].astype(np.float64).copy(); a=g["action"][i].astype(np.float64).copy()
        obs=np.zeros(task_s.obs_dim); rew=ctypes.c_double(0); suc=ctypes.c_ubyte(0); st=ctypes.c_int(0)
        L.emu_adroit_step(ctypes.c_void_p(h), ctypes.byref(t), p(qp),p(qv),p(qa),p(sh),p(tg),p(a),p(am.copy()),p(ar.copy()),p(obs),ctypes.byref(rew),ctypes.byref(suc),ctypes.byref(st),ctypes.c_int(0))
        E.append(np.abs(obs-g["obs"][i]).max())
    E=np.array(E)
    print('  outliers:', [(int(3*k), float('%.1e'%E[k]), int(g['ncon'][3*k]), int(g['nefc'][3*k]), int(g['noslip_iter'][3*k])) for k in np.nonzero(E>1e-6)[0]])
    print(task,'fp64 emulator: p50 %.1e p90 %.1e p99 %.1e max %.1e; n>1e-6: %d of %d'%(np.median(E),np.quantile(E,.9),np.quantile(E,.99),E.max(),(E>1e-6).sum(),len(E)))
