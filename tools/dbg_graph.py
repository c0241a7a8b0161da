import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import make_model
from oracle import net_oracle
m,_=make_model("n",9,dtype="f16"); m=m.to("cuda")
x=net_oracle.synth_image(1,3,64,64,9).cuda()
m.model.use_graph=True
o=m(x); torch.cuda.synchronize(); print("ok", float(o['semi'].abs().max()))
