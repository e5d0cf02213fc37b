"""PinSAGE item-item path — reference: pinsage/{sampler,layers,model}.py."""
