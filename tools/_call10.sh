cd $GRAFT_REPO_ROOT
python tools/_dbg_recycling.py
echo "--- chain off"
KRYPY_AMD_MGS_CHAIN=0 python tools/_dbg_recycling.py
