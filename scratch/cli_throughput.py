"""CLI throughput: 40 synthetic 256 x 256 images, FFHQ-architecture seeded weights, gaussian_blur, Heun-30, batch 8."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import PIL.Image, torch
from bench import smooth_images
import generate_conditional as gc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tmp = tempfile.mkdtemp(); data = os.path.join(tmp, "data"); os.makedirs(data)
for i, im in enumerate(smooth_images(n, 256, 7)):
    PIL.Image.fromarray(im.permute(1, 2, 0).numpy(), "RGB").save(os.path.join(data, f"img{i:08d}.png"))
out = os.path.join(tmp, "out")
os.environ["FH_PHASE_TIMES"] = "1"
t0 = time.time()
gc.main([f"--outdir={out}", f"--dataset_path={data}", "--synthetic_weights=ffhq", "--num_steps=30", f"--total_images={n}",
         "--max_batch_size=8", "--operator_name=gaussian_blur", "--solver=heun", "--conditioning_mechanism=online_covariance",
         "--image_base_covariance=dct_diagonal"])
torch.cuda.synchronize()
dt = time.time() - t0
print(f"CLI: {n} images in {dt:.1f} s = {n / dt:.3f} images/s (includes model build, first-batch compile / caches, PNG + metrics)")
print(open(os.path.join(out, "results.txt")).read()[-400:])
