#!/usr/bin/env python3
"""Per-launch fabric traffic of one steady-state step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_r2f.sh) next
to the algorithmic bytes of the layer: explains the dominant kernel's `roofline.traffic` (256 MB average per launch against
~67 MB algorithmic).  usage: traffic_per_layer.py gpurun_out/r2f_prof_cfg1 > profiles/traffic_r02_per_layer_cfg1.md"""
import csv, glob, sys
root = sys.argv[1].rstrip("/") + "/"
def load(name):
    f = glob.glob(root + f"pmc_{name}/*/*counter_collection.csv")[0]
    return sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
idx = [i for i, r in enumerate(fe) if "rowwin" in r["Kernel_Name"]]
per, s0 = idx[1] - idx[0], idx[-1]
B = 8
def mb(*a):
    x = 4.0
    for v in a:
        x *= v
    return x / 1e6
# algorithmic MB at B=8 512x512x27: (input read once, weights, output written once)
LAY = {"conv1": (mb(B, 512, 512, 27), 0.34, mb(B, 256, 256, 64)), "conv2": (mb(B, 256, 256, 64), 0.8, mb(B, 128, 128, 128)),
       "conv3": (mb(B, 128, 128, 128), 3.3, mb(B, 64, 64, 256)), "conv4": (mb(B, 64, 64, 256), 4.7, mb(B, 32, 32, 512)),
       "conv5": (mb(B, 32, 32, 512), 9.4, mb(B, 16, 16, 512)), "conv6": (mb(B, 16, 16, 512), 18.9, mb(B, 8, 8, 1024)),
       "deconv5": (mb(B, 8, 8, 1024), 33.6, mb(B, 16, 16, 512)), "deconv4": (mb(B, 16, 16, 1026), 16.8, mb(B, 32, 32, 256)),
       "deconv3": (mb(B, 32, 32, 770), 6.3, mb(B, 64, 64, 128))}
order = ["conv2", "conv3", "conv4", "conv5", "conv6", "deconv5", "deconv4", "deconv3"]
print("# Fabric traffic per launch of one step, B=8 512x512x27 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each)\n")
print("FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streams (uncalibrated for the narrower accesses of the "
      "small kernels).  It counts the L2s' memory-side requests: Infinity-Cache hits are included, so this is fabric traffic, an upper bound "
      "of HBM traffic.\n")
print("| launch | read MB | written MB | layer | algorithmic read (input + weights) MB | algorithmic write MB | read ratio |")
print("|---|---:|---:|---|---:|---:|---:|")
k = 0; tot = [0.0, 0.0, 0.0, 0.0, 0]
for r, w in zip(fe[s0:s0 + per], wr[s0:s0 + per]):
    n = r["Kernel_Name"].replace("void ", "").replace("vstab::", "").split("(")[0]
    f = float(r["Counter_Value"]) * 1024 * 2 / 1e6; ww = float(w["Counter_Value"]) * 1024 / 1e6
    L = None
    if n.startswith("conv_mfma_kernel<128, 128"):
        L = order[k]; k += 1
        tot[0] += f; tot[1] += ww; tot[2] += LAY[L][0] + LAY[L][1]; tot[3] += LAY[L][2]; tot[4] += 1
    elif "rowwin" in n:
        L = "conv1"
    if L:
        a = LAY[L]
        print(f"| `{n}` | {f:.1f} | {ww:.1f} | {L} | {a[0] + a[1]:.1f} | {a[2]:.1f} | {f / (a[0] + a[1]):.2f} |")
    else:
        print(f"| `{n[:60]}` | {f:.1f} | {ww:.1f} | | | | |")
print(f"\nThe dominant kernel `conv_mfma_kernel<128, 128, 2, 2, true, true>` ({tot[4]} launches): {tot[0] / tot[4]:.0f} MB read + {tot[1] / tot[4]:.0f} MB written per "
      f"launch on average, against {tot[2] / tot[4]:.0f} + {tot[3] / tot[4]:.0f} MB algorithmic.")
