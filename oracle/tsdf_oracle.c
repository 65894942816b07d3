int tsdf_oracle_stub(void){return 0;}
