"""CPU oracle for the Ray3D lifting path - TEST INFRASTRUCTURE ONLY (see ray3d_oracle.h)."""
