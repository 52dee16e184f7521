"""profiles/<round>/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh.
usage: python tools/make_traffic.py gpurun_out/prof_<soft tag> gpurun_out/prof_<rigid tag> > profiles/<round>/traffic.json"""
import csv, glob, json, os, sys

def mean_kb(root, sub, counter, steps_per_launch):
    """mean over the step launches, per STEP of all environments (a launch runs steps_per_launch consecutive steps)"""
    vals = []
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].replace(" ", "")
                # step launches: the split kernel, or MODE 0 (template argument before the MULTI flag) of the 16-lane / round-1 kernel templates
                head = name.split(">(")[0]
                is_step = "usim_step32_kernel" in name or (("usim_step16_kernel" in name) and (head.endswith(",0,true") or head.endswith(",0,false") or head.endswith(",0"))) \
                    or ("usim_step_kernel" in name and head.endswith(",0"))
                if is_step and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
    if not vals:
        return None, 0
    med = sorted(vals)[len(vals) // 2]             # launches of the full steps_per_launch steps (a shorter tail launch, a cold first launch are left out)
    sel = [v for v in vals if 0.5 * med < v < 2.0 * med]
    return sum(sel) / len(sel) / steps_per_launch, len(sel)

spl = int(sys.argv[3]) if len(sys.argv) > 3 else 256
out = {}
for tag, root in (("soft", sys.argv[1]), ("rigid", sys.argv[2])):
    f, nf = mean_kb(root, "pmc_fetch", "FETCH_SIZE", spl)
    w, nw = mean_kb(root, "pmc_write", "WRITE_SIZE", spl)
    out[tag] = {"fetch_kb": f, "fetch_kb_x2": None if f is None else 2 * f, "fetch_kb_launches": nf, "write_kb": w, "write_kb_launches": nw, "steps_per_launch": spl}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), separate passes (tools/profile.sh), mean over the full step launches of "
               "`bench.py --steps 2048 --warmup 256` at 4096 envs, divided by the steps per launch: KB per step of all environments. "
               "fetch_kb is the raw counter; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane streaming reads. Since round 2 "
               "the state is environment-major and the scalar words are read as 16-byte quads, the lattice words 4 bytes per lane over 64-byte runs: "
               "the correction applies to the former, is uncalibrated for the latter -- fetch_kb_x2 is the upper bound, and the figure to compare "
               "with the algorithmic bytes. WRITE_SIZE is uncalibrated (reported raw). bench.py reports fetch_kb_x2 + write_kb.")
print(json.dumps(out, indent=1))


def counter_mean(root, sub, counter):
    vals = []
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "usim_step" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
    if not vals:
        return None
    med = sorted(vals)[len(vals) // 2]
    sel = [v for v in vals if 0.5 * med < v < 2.0 * med]
    return sum(sel) / len(sel)


# issue.json (4th argument: its path): instruction-issue figures of the step kernel per launch, from the SQ passes of tools/profile.sh
if len(sys.argv) > 4:
    iss = {}
    for tag, root, waves in (("soft", sys.argv[1], 2048), ("rigid", sys.argv[2], 1024)):
        valu, wc = counter_mean(root, "pmc_sq1", "SQ_INSTS_VALU"), counter_mean(root, "pmc_sq1", "SQ_WAVE_CYCLES")
        act, lds, conf = counter_mean(root, "pmc_sq2", "SQ_ACTIVE_INST_VALU"), counter_mean(root, "pmc_sq2", "SQ_ACTIVE_INST_LDS"), counter_mean(root, "pmc_sq2", "SQ_LDS_BANK_CONFLICT")
        if valu and wc:
            iss[tag] = {"valu_inst_per_wave_step": valu / (waves * spl), "valu_active_share_of_wave_cycles": None if not act else act / wc,
                        "lds_bank_conflict_cycles_per_lds_active_cycle": None if not (lds and conf) else conf / lds, "waves": waves, "steps_per_launch": spl,
                        "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES / SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT (tools/profile.sh), mean over the full step launches"}
    open(sys.argv[4], "w").write(json.dumps(iss, indent=1))
