python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | grep -v amdgpu.ids | tail -2
python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15
