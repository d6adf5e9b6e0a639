cd $GRAFT_REPO_ROOT
for p in 1 2 3; do echo "process $p"; timeout -s KILL 200 python tools/bimodal_probe.py 2>&1 | tail -5; done
