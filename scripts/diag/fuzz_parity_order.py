ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
         "map.linear.weight", "map.linear.bias"]
