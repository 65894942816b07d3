"""Seeded synthetic inputs of SURVEY.md section 8(d) -- keyframe windows (scene.make_window), depth scans
(scene.make_scans), tracking pairs, and the analytic-room frame sequence of BASELINE configs[3] (room).  Shared by the
tests, bench.py and the tools; neither product code (tandem_amd/ never imports it) nor part of the oracle."""
