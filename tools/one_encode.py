import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate_torch
T=4; n=(1<<30)//4
src=generate_torch(sys.argv[1] if len(sys.argv)>1 else "rand12",T,n,42)
st=Stenos(1)
dst=torch.empty(st.bound(src.numel()),dtype=torch.uint8,device="cuda")
try: st.compress(src,T,dst)
except Exception as e: pass
