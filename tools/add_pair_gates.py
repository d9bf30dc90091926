"""Attach the joint-box gates of the hull pairs (gymnasium_robotics_amd/mjcf/pair_gates.py) to packaged model blobs without recompiling them from MJCF:
    python tools/add_pair_gates.py gymnasium_robotics_amd/models/kitchen.npz [...]
(tools/compile_models.py requests the gates for the Fetch and kitchen models when it compiles them: capacity["pair_gates"])"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd.mjcf.compiler import load_model, save_model
from gymnasium_robotics_amd.mjcf.pair_gates import compute_pair_gates

for path in sys.argv[1:]:
    m = load_model(path)
    t0 = time.time()
    T = dict(m.tables)
    T["devpair_gate"], T["gate_qadr"], T["gate_box"], rep = compute_pair_gates(T)
    m.tables.update(devpair_gate=T["devpair_gate"], gate_qadr=T["gate_qadr"], gate_box=T["gate_box"])
    m.info["pair_gates"] = [dict(pair=p_, geoms=[g1_, g2_], joints=j_, box=b_, slack=s_) for p_, g1_, g2_, j_, b_, s_ in rep]
    save_model(m, path)
    proven = [r for r in rep if r[4] is not None]
    print(f"{path}: {len(proven)} gates for {len(rep)} analysed pairs of {len(T['devpair'])} candidates, {time.time() - t0:.0f} s")
