"""Print the kernel timeline of the LAST FactorGraph.update before the NeuS bench in a rocprofv3
--kernel-trace CSV of bench.py (start offset, gap to previous kernel, duration, grid, name)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
neus = next((i for i, r in enumerate(rows) if 'neus_' in r['Kernel_Name']), len(rows))
reps = [i for i, r in enumerate(rows[:neus]) if 'reproject_kernel' in r['Kernel_Name']]
lo = reps[-1]
t0 = int(rows[lo]['Start_Timestamp']); prev = t0
for r in rows[lo:neus]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'_ZN2ck.*?(kernel_[a-z_0-9]+).*', r'ck::\1', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    n = re.sub(r'at::native::', '', n)[:64]
    print(f"{(s-t0)/1e3:8.1f} gap{(s-prev)/1e3:6.1f} dur{(e-s)/1e3:7.1f} grid{r['Grid_Size_X']:>9s} {n}")
    prev = e
    if 'cvx_upsample' in n:
        break
