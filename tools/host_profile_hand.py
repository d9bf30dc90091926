import cProfile, pstats, os, sys, io
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
w="hand_touch"; W=bench.WORKLOADS[w]; n=W["worlds"]
env=bench.make_env(w,n,"cuda:0",0); env.reset(seed=0)
bench._set_elapsed(env, np.arange(n) % env.max_episode_steps)
g=torch.Generator(device="cuda:0"); g.manual_seed(1)
for _ in range(20): env.step(torch.rand(n,20,device="cuda:0",generator=g)*2-1)
torch.cuda.synchronize()
env.chain_events=[]; env.kernel_events=[]
pr=cProfile.Profile(); pr.enable()
for _ in range(40): env.step(torch.rand(n,20,device="cuda:0",generator=g)*2-1)
pr.disable(); torch.cuda.synchronize()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])

ke=env.kernel_events
print("step kernel ms:", " ".join(f"{a.elapsed_time(b):.1f}" for a,b in ke[:12]))
for st,due,k,e0,e1 in env.chain_events[:14]:
    # position of the chain's start / end relative to the start of the step kernel of the step it was started in (kernel_events index = st - first)
    first=env.chain_events[0][0]
    i=st-first
    if i < len(ke):
        print(f"chain started in step {st} due {due} ({k} worlds): device time {e0.elapsed_time(e1):.1f} ms; starts {ke[i][0].elapsed_time(e0):.1f} ms after that step's kernel started, ends {ke[i][0].elapsed_time(e1):.1f} ms after it")
