"""One launch of the chain kernel on a named scene / type for profiler passes: config_once.py <scene> <type> [res] [spp]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene, typ = sys.argv[1], sys.argv[2]
res = int(sys.argv[3]) if len(sys.argv) > 3 else 256
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 256
sd = pkg.scenes.triangle_soup(int(os.environ.get('N_TRIS', 2000)), res) if scene == 'triangle_soup' else pkg.scenes.SCENES[scene](res=res)
cfg = pkg.abi.make_config(type=typ, max_depth=8, direct_samples=-1, work_units=65536, luminance_samples=655360, sample_count=spp)
c = pkg.Context(cfg, sd)
c.seed(0x5EED)
c.run(res * res * spp)
st = c.stats()
print("mutations %d rays/mut %.2f evals/mut %.2f kernel ms %s" % (st.mutations, st.rays / st.mutations, st.path_evals / st.mutations, c.kernel_time()))
print({k: round(v, 4) if v is not None else None for k, v in st.ratios().items()})
