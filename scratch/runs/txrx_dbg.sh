cd $GRAFT_REPO_ROOT
python scratch/txrx_dbg.py host256 2>&1 | tail -4
python scratch/txrx_dbg.py bulk 2>&1 | tail -3
python scratch/txrx_dbg.py 1048576 2>&1 | tail -3
MCRX_SCOUT_ROUNDS=1 python scratch/txrx_dbg.py 1048576 2>&1 | tail -3
MCRX_NO_SPEC=1 python scratch/txrx_dbg.py 1048576 2>&1 | tail -3
MCRX_SERIAL=1 python scratch/txrx_dbg.py 1048576 2>&1 | tail -3
