"""TEST INFRASTRUCTURE ONLY (oracle shim)."""
