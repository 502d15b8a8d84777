import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate_torch
T=4; n=(2<<30)//4
src=generate_torch("rand12",T,n,42)
st=Stenos(1); st.set_profiling(True)
dst=torch.empty(st.bound(src.numel()),dtype=torch.uint8,device="cuda")
os.environ["STENOS_DEBUG_PHASES"]="1"
for lds in ("6320","4096","3328","8192","12288"):
    os.environ["STENOS_EXP_SMALL_LDS"]=lds
    ms=[]
    for i in range(4):
        try: c=st.compress(src,T,dst)
        except Exception as e: pass
        ms.append(st.kernel_ms(0))
    print("lds",lds,"no-LZ encode_blocks ms (2 GiB):",["%.3f"%m for m in ms])
