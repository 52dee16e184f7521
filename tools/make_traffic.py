"""profiles/<round>/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh.
usage: python tools/make_traffic.py gpurun_out/prof_<soft tag> gpurun_out/prof_<rigid tag> > profiles/<round>/traffic.json"""
import csv, glob, json, os, sys

def mean_kb(root, sub, counter):
    vals = []
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].replace(" ", "")
                # step launches: the split kernel, or MODE 0 (last template argument) of the 16-lane / round-1 kernel templates
                is_step = "usim_step32_kernel" in name or (("usim_step16_kernel" in name or "usim_step_kernel" in name) and name.split(">(")[0].endswith(",0"))
                if is_step and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)

out = {}
for tag, root in (("soft", sys.argv[1]), ("rigid", sys.argv[2])):
    f, nf = mean_kb(root, "pmc_fetch", "FETCH_SIZE")
    w, nw = mean_kb(root, "pmc_write", "WRITE_SIZE")
    out[tag] = {"fetch_kb": f, "fetch_kb_launches": nf, "write_kb": w, "write_kb_launches": nw}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), separate passes (tools/profile.sh), mean per launch of the step kernel of "
               "`bench.py --steps 500 --warmup 50` at 4096 envs. Raw counter values: the gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md is "
               "calibrated for 16 B/lane streaming reads; this kernel reads 4 B/lane rows, so the uncorrected value is reported and 2x it is the upper bound.")
print(json.dumps(out, indent=1))
