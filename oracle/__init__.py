"""CPU oracle for the Point-Teacher hot path.  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker;
never by the product package (point_teacher_amd)."""
