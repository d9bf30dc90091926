"""Do the settle chains of a hand-manipulate reset overlap with the step launch?  Times the step kernel of 16 384 worlds, ten dependent launches of 164 worlds
on a second stream, and both together (GPU only).  Result of round 2: profiles/overlap_probe_r02.txt"""
import sys, time, ctypes
import os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import numpy as np, torch
import gymnasium_robotics_amd as grx
from gymnasium_robotics_amd import _native
n=16384
env=grx.make_vec("HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0)
g=torch.Generator(device="cuda:0"); g.manual_seed(0)
a=torch.rand(n,20,device="cuda:0",generator=g)*2-1
for _ in range(3): env.step(a)
torch.cuda.synchronize()
L=env._L
def main_launch():
    _native.check(L.grx_hand_step(env._h, ctypes.byref(env.task), ctypes.byref(env._bufs), n, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
ar=env._arena(); k=164
ar["qpos"][:k]=env.qpos[:k]; ar["qvel"][:k]=0
side=env._side[0]; ar_bufs=env._arena_bufs(0)
def side_chain(m=10):
    sp=ctypes.c_void_p(side.cuda_stream)
    for _ in range(m): _native.check(L.grx_hand_step(env._h, ctypes.byref(env.task), ctypes.byref(ar_bufs), k, 0, sp))
def T(f, reps=5):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e3
print("main only %.2f ms"%T(main_launch))
print("side chain only (10 launches of %d worlds) %.2f ms"%(k,T(side_chain)))
print("side 1 launch %.2f ms"%T(lambda: side_chain(1)))
def both():
    side_chain(); main_launch()
print("both (side first) %.2f ms"%T(both))
def both2():
    main_launch(); side_chain()
print("both (main first) %.2f ms"%T(both2))
# masked launch cost
env.mask.zero_(); env.mask[:k]=1
def masked():
    for _ in range(10): _native.check(L.grx_hand_step(env._h, ctypes.byref(env.task), ctypes.byref(env._bufs_masked), n, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
print("10 masked full-grid launches %.2f ms"%T(masked))
