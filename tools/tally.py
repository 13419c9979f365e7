import sys, os
sys.path.insert(0, '/root/repo')
os.environ["RT_MI355_DEBUG_TALLIES"]="1"
import bench, argparse
from raytracer_amd import _ffi as F
from raytracer_amd.renderer import Renderer
args = argparse.Namespace(width=1920,height=1080,spp=4,bounces=60,detail=4)
scene = bench.load_scene(args)
r = Renderer(0); r.upload(scene)
r.set_option(F.RT_OPT_COUNT_TRAVERSAL,1)
r.reset_stats()
r.render_sums(scene.params,0,4)
print(r.stats())
r.reset_stats()
