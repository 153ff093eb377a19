"""CPU parity oracle for the CrowdNav++ hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (crowdnav_prediction_attngraph_amd) never does.
"""
