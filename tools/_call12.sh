cd $GRAFT_REPO_ROOT
for pf in 1 0; do echo "PF=$pf"; KRYPY_AMD_CHAIN_PF=$pf python tools/_dbg_arnoldi2.py 2>&1 | tail -8; done
echo chain off; KRYPY_AMD_MGS_CHAIN=0 python tools/_dbg_arnoldi2.py 2>&1 | tail -5
