import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate_torch
T=4; n=(2<<30)//4
src=generate_torch("rand12",T,n,42)
st=Stenos(1); st.set_profiling(True)
dst=torch.empty(st.bound(src.numel()),dtype=torch.uint8,device="cuda")
back=torch.empty_like(src)
for w in sys.argv[1:] or ["default"]:
    if w=="default": os.environ.pop("STENOS_WAVES_PER_CU",None)
    else: os.environ["STENOS_WAVES_PER_CU"]=w
    for dbg in (0,7):
        os.environ["STENOS_DEBUG_PHASES"]=str(dbg)
        ms=[]
        for i in range(4):
            try: c=st.compress(src,T,dst)
            except Exception as e: pass
            ms.append(st.kernel_ms(0))
        print("waves/CU",w,"dbg",dbg,"encode_blocks ms (2 GiB):",["%.3f"%m for m in ms])
    os.environ["STENOS_DEBUG_PHASES"]="0"
    c=st.compress(src,T,dst); idx,_=st.last_index(); st.decompress(dst,T,c,back,index_ptr=idx); print("roundtrip ok", bool(torch.equal(back,src)), c)
