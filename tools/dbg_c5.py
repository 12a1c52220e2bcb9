import sys; sys.path.insert(0,'/root/repo')
import numpy as np
import pop_up_slam_amd as P
from pop_up_slam_amd import pipeline
frames = pipeline.popup_sequence(1000)
pl, g, pp, stats = pipeline.gpu_pipeline(step=2)
for k, fr in enumerate(frames):
    try:
        pl.process(fr)
    except P.PpsError as e:
        st=g.stats(); print("frame",k,"err",e,"k%5",k%5,"n ids",len(fr.ids), "new landmarks?", st["n_planes"], "fronts", st["n_fronts"], "max_front", st["max_front"])
        # which landmarks have a single observation
        break
else:
    print("ok all")
