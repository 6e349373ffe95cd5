"""funscript_flow_amd -- MI355X-native (gfx950) implementation of Funscript-Flow's per-frame-pair
motion path (Farneback flow -> |div| argmax -> radial weighted mean) behind the reference's backend
switch (FunscriptFlow.pyw:854-873).  See DESIGN.md / INTEGRATION.md."""
