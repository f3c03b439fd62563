"""host seconds spent in the pieces of a stage boundary of generate() (config C2's geometry by default): which host call makes the device wait
between the last Euler step of a stage and the re-noising kernel of the next?  argv: [c2 | c3]"""
import os, sys, time, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
import bench
from pyflow_hip import ops
w = sys.argv[1] if len(sys.argv) > 1 else "c2"
H, W, temp, steps1, stepsv = bench.WORKLOADS["c2_384p_121f" if w == "c2" else "c3_768p_241f"]
pipe, dcfg, dsd = bench.build_pipeline("cuda:0")
emb = bench.synthetic_prompt(dcfg, "cuda:0")
acc = collections.defaultdict(lambda: [0, 0.0])


def timed(obj, name, label=None):
    f = getattr(obj, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        e = acc[label or name]
        e[0] += 1
        e[1] += time.perf_counter() - t0
        return r
    setattr(obj, name, wrap)


timed(pipe, "sample_block_noise")
timed(pipe, "_to_device_async")
timed(pipe.scheduler, "set_timesteps")
timed(pipe, "_plan")
timed(pipe, "_history")
timed(pipe, "_pyramid")
timed(pipe.dit, "forward_tokens")
timed(ops, "renoise_upsample")
timed(ops, "cfg_euler_step")
for i in range(2):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.generate(prompt_embeds=emb, height=H, width=W, temp=temp, num_inference_steps=steps1, video_num_inference_steps=stepsv,
                  guidance_scale=7.0, video_guidance_scale=5.0, generator=torch.Generator().manual_seed(i), output_type="latent", save_memory=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"video {i}: host returned after {t1 - t0:.3f} s, device done after {t2 - t0:.3f} s")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:22s} calls {n:5d}  host total {t * 1e3:9.1f} ms  mean {t / n * 1e3:8.3f} ms")
