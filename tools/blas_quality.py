import importlib, sys, os, numpy as np, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")
import oracle_py as O
from test_gpu_parity import _download_tree
def rays(n, seed, R):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n,3)); o = (o/np.linalg.norm(o,axis=1,keepdims=True)*R).astype(np.float32)
    t = rng.uniform(-0.05,0.05,size=(n,3)).astype(np.float32)
    return o, (t-o).astype(np.float32)
for subdiv in (2,3):
    shape = hrt.scenes._blob_shape(subdiv, 0.06, 9)
    scene = {"name":"one", "instances":[hrt.scenes._tri_instance(shape, hrt.scenes.WHITE)], "camera": hrt.scenes._soup_camera(), "background": hrt.scenes.BACKGROUND.copy(), "width":64,"height":64,"spp":1}
    o,d = rays(20000, 1, 0.5)
    for build in ("device","host"):
        os.environ["HRT_BUILD"] = build
        r = hrt.Renderer(0,0); r.load_scene(scene)
        nodes, prims = _download_tree(hrt, r)
        res = O.bvh8_trace(nodes.ctypes.data, prims.ctypes.data, o, d)
        hit = (res[3] != 0xffffffff).mean()
        print(f"subdiv {subdiv} ({len(shape)} tris) {build}: nodes {len(nodes)//80}, node visits/ray {res[5]/len(o):.2f}, prim tests/ray {res[6]/len(o):.2f}, hit frac {hit:.2f}", flush=True)
        r.close()
