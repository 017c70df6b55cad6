"""BVH traversal statistics of the 2000-triangle soup: node visits / primitive tests per ray (scene part of the algorithmic bytes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sd = pkg.scenes.triangle_soup(n, 256)
cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=65536, luminance_samples=100000, sample_count=16)
c = pkg.Context(cfg, sd); c.seed(1)
c.run(65536 * 4)
t = time.time(); c.run(65536 * 16); dt = time.time() - t
st = c.stats()
print("tris %d: %.3e mut/s; rays/mut %.2f; node visits/ray %.1f prim tests/ray %.1f; bytes/mutation from the scene %.0f" % (
    n, 65536 * 16 / dt, st.rays / st.mutations, st.bvh_node_visits / st.rays, st.bvh_prim_tests / st.rays,
    (st.bvh_node_visits * 128 + st.bvh_prim_tests * 64) / st.mutations))
