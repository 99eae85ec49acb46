"""ORACLE package: CPU restatement of the reference hot path.  Test infrastructure only."""
