cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -4
timeout 600 python -m pytest tests/test_spectrogram.py -m gpu -q 2>&1 | grep -v amdgpu | tail -2
timeout 600 python tools/e2e_train_throughput.py --from-audio --steps 100 2>&1 | grep -v amdgpu | tail -4
