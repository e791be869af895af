import importlib.util, os, sys, tempfile
import numpy as np
REPO="/root/repo"
spec = importlib.util.spec_from_file_location("beatrice_vst_amd", os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
bv = importlib.util.module_from_spec(spec); sys.modules["beatrice_vst_amd"]=bv; spec.loader.exec_module(bv)
sys.path.insert(0, os.path.join(REPO,"tools")); import make_model
d=tempfile.mkdtemp(); make_model.make_model(d, n_speakers=3)
x = bv.synth_audio(160*40, seed=3)
res={}
for name, path in (("o", os.path.join(REPO,"oracle","libbeatrice_oracle.so")), ("p", bv.PRODUCT_LIB), ("p2", bv.PRODUCT_LIB)):
    abi=bv.Abi(path); m=bv.Models(abi,d); pc=abi.CreatePhoneContext1()
    outs=[]
    for i in range(40):
        out=np.zeros(128,np.float32)
        abi.ExtractPhone1(m.phone, bv.fptr(np.ascontiguousarray(x[i*160:(i+1)*160])), bv.fptr(out), pc)
        outs.append(out)
    abi.DestroyPhoneContext1(pc); m.close(); res[name]=np.array(outs)
for k in ("p","p2"):
    d_=np.abs(res["o"]-res[k]).max(axis=1)
    print(k, "per-hop max-abs:", np.round(d_,4).tolist())
    bad=np.argwhere(np.abs(res["o"]-res[k])>1e-6)
    print(k, "n bad elems", len(bad), "first", bad[:10].tolist())
