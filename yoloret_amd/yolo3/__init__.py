"""Host-side mirror of the reference's ``yolo3`` package for the detection forward path."""
