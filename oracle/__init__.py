"""CPU oracle for the Krylov hot path.  Test infrastructure only - see krylov_ref.py."""
