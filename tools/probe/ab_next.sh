python tools/probe/bt_clk.py
