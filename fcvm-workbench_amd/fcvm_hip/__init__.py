"""fcvm_hip: host binding of libfcvm_hip.so (MI355X hot path of the fcVM tet10 elastoplastic solver)."""
