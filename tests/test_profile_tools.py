"""tools/trace_one_sort.py on a hand-made rocprofv3 kernel trace: the last sort is cut out at its first kernel, gaps and
totals are what the timestamps say."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_one_sort_cuts_out_the_last_sort(tmp_path):
    d = tmp_path / "run" / "box"
    d.mkdir(parents=True)
    rows = [("dq::text_hist_kernel(a)", 1000, 2000, 512),            # an earlier sort
            ("void dq::radix_rank_kernel<int>(b)", 3000, 9000, 8192),
            ("dq::text_hist_kernel(a)", 20000, 30000, 512),           # the last one
            ("void dq::radix_rank_kernel<int>(b)", 35000, 44000, 8192),
            ("void dq::seg_fused_kernel<int, true>(c)", 50000, 76000, 4096)]
    with open(d / "1_kernel_trace.csv", "w") as f:
        f.write("Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size_X\n")
        for name, s, e, g in reversed(rows):                          # (the tool sorts by start time itself)
            f.write(f'"{name}",{s},{e},{g}\n')
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_one_sort.py"), str(tmp_path / "run")],
                       capture_output=True, text=True, check=True)
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 4 and "text_hist_kernel" in lines[0] and "seg_fused_kernel" in lines[2]
    assert "gap    5.0" in lines[1] and "gap    6.0" in lines[2]          # 35 - 30 us, 50 - 44 us
    assert lines[3] == "total 56.0 us, kernels 45.0 us, 3 launches"
