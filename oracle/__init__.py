"""CPU oracle for the KKT factor+solve hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (ipopt_amd/, include/) never does."""
