"""Host-side mirrors of the reference helpers around the SCONE hot path (macarons/utility/*): thin Python over macarons_amd.ops."""
