cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -k "ctc or lattice or repair or golden or loss or gradients" 2>&1 | grep -v amdgpu | tail -4
timeout 300 python tools/ctc_time.py 2>&1 | grep -v amdgpu | tail -6
timeout 300 python tools/ctc_time.py --batch 8 --frames 4000 2>&1 | grep -v amdgpu | tail -6
timeout 600 python bench.py --no-cpu-baseline --no-also | cut -c1-400
