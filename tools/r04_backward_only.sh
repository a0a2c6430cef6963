#!/bin/bash
# the training-kernel part of tools/r04_evidence.sh alone (after a change to csrc/train.hip): suite, bench, kernel trace
out=gpurun_out/${1:-r04bw}; mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_gpu_backward.py -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
timeout 200 python tools/bench_backward.py > $out/bench_backward.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace_backward -o p -- python tools/bench_backward.py > $out/trace_backward.log 2>&1
python - "$out" <<'PY'
import os, sys
sys.argv = ["x", sys.argv[1]]
src = sys.argv[1]
sys.path.insert(0, "tools")
import importlib.util
spec = importlib.util.spec_from_file_location("summ", "tools/r04_summarise.py")
# only the pieces needed: reuse trace_table / short by exec'ing the helper definitions
code = open("tools/r04_summarise.py").read().split("md = [")[0]
ns = {"__name__": "helpers"}
sys.argv = ["r04_summarise.py", src, "--helpers-only"]
exec(compile(code.replace('if "--power-only" in sys.argv:', 'if False:'), "helpers", "exec"), ns)
tk = ["# r04: the training kernels (csrc/train.hip, SURVEY 8(f)4) on the round's build\n", "## tools/bench_backward.py\n", "```"]
tk += [l for l in open(os.path.join(src, "bench_backward.txt")).read().strip().splitlines() if "amdgpu.ids" not in l] + ["```\n"]
tk.append("## rocprofv3 --kernel-trace --stats of the same command\n")
ns["trace_table"](os.path.join(src, "trace_backward", "p_results.db"), tk)
os.makedirs(os.path.join(src, "summary"), exist_ok=True)
open(os.path.join(src, "summary", "r04_training_kernels.md"), "w").write("\n".join(tk) + "\n")
print("\n".join(tk))
PY
rm -rf $out/trace_backward
