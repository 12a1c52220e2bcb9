"""Eight handles solved concurrently on eight host threads (the bench's concurrent_graphs_one_gpu section on its own)."""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pop_up_slam_amd as P
from pop_up_slam_amd import synth
hs = []
for k in range(8):
    g = P.Graph(); synth.corridor(300, 60, seed=k).replay(g); g.save_state(); g.batch_optimize(); hs.append(g)
def work(k):
    for _ in range(4):
        hs[k].restore_state(); hs[k].batch_optimize()
th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
for t in th: t.start()
for t in th: t.join()
print("threads ok", [h.stats()["lm_iterations"] for h in hs])
