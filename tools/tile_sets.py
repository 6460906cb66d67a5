# conv order in adaf_resnet50: stem, then per block conv1, conv2, conv3, (downsample)
import sys
mode = sys.argv[1]
tiles = [0]
cfg = [(64,3),(128,4),(256,6),(512,3)]
for pl, nb in cfg:
    for b in range(nb):
        outs = [pl, pl, pl*4] + ([pl*4] if b == 0 else [])
        for co in outs:
            if mode == "big":
                tiles.append(31 if co >= 128 else 32)
            elif mode == "mid":
                tiles.append(32)
            elif mode == "small":
                tiles.append(33)
            else:
                tiles.append(0)
print(",".join(map(str, tiles)))
