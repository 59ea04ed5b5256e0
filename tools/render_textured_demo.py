import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import raytracer_amd as rt
from test_textures import TEXTURED_SCENE, WITH_JPEG
from oracle_api import ASSETS
os.makedirs('gpurun_out/tex', exist_ok=True)
open('gpurun_out/tex/scene.json','w').write(TEXTURED_SCENE % WITH_JPEG)
ds = rt.DeviceScene(rt.Scene(os.path.abspath('gpurun_out/tex/scene.json'), 640, 400, 3, base_dir=ASSETS))
rgb,_,_,st = ds.render_host(want_ids=False, flags=rt.RT_RENDER_TIME)
rt.write_png('gpurun_out/tex/textured.png', rgb)
print(st['kernel_ms'])
