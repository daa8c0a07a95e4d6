"""TEST INFRASTRUCTURE ONLY (oracle shim) for the un-vendored torch_geometric."""
